"""Phase cycles of the streaming weight-gradient GEMM's chunk loop from a -DGCP_TN_TIMING build of tn_gemm.hip (tn_stream_kernel):
    GCPNET_HIP_LIB=<timing .so> python tools/tn_stream_phases.py [rows M N]...
Per wave, average s_memtime ticks per 16-row chunk of: wait + barrier | split pass + barrier | DMA issue | products."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gcpnet_amd import _lib, ops  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(159913, 128, 144), (999995, 256, 284)]
lib = _lib.load()
for rows, M, N in shapes:
    a = torch.randn(rows, M, device="cuda")
    b = torch.randn(rows, N, device="cuda")
    for _ in range(3):
        out = ops._tn_weight_grad(a, b)
    torch.cuda.synchronize()
    splits = lib.gcpnet_tn_splits(rows, M, N)
    chunks = (rows + 15) // 16 / splits
    waves = 8 if (M > 128 or N > 160) else 4
    t = out[0, :4 * waves].cpu().double().reshape(waves, 4) / splits
    print(f"rows {rows} M {M} N {N}: {splits} splits, {chunks:.1f} chunks each; s_memtime ticks per chunk")
    for w in range(waves):
        print(f"  wave {w}: wait+split {t[w, 0] / chunks:8.1f}  load issue {t[w, 1] / chunks:8.1f}  barrier {t[w, 2] / chunks:8.1f}  products {t[w, 3] / chunks:8.1f}   total {t[w].sum():10.0f}")
