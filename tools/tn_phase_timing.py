"""Phase cycles of the weight-gradient GEMM's chunk loop from a -DGCP_TN_TIMING build (see tn_gemm.hip):
    hipcc ... -DGCP_TN_TIMING -c gcpnet_amd/csrc/tn_gemm.hip -o /tmp/tn_t.o; link with the other objects into a second .so;
    GCPNET_HIP_LIB=<that .so> python tools/tn_phase_timing.py [rows M N]...
Prints, per wave, the average cycles per 32-row chunk of: wait + barrier | DMA issue | products | rest."""
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
    chunks = (rows + 31) // 32 / splits
    waves = 8 if (M > 128 or N > 160 or os.environ.get("GCPNET_TN_EIGHT_WAVES")) else 4
    t = out[0, :4 * waves].cpu().double().reshape(waves, 4) / splits
    print(f"rows {rows} M {M} N {N}: {splits} splits, {chunks:.1f} chunks each; cycles per chunk (100 MHz s_memtime ticks x 24 if constant-rate)")
    for w in range(waves):
        print(f"  wave {w}: wait {t[w, 0] / chunks:8.1f}  issue {t[w, 1] / chunks:8.1f}  products {t[w, 2] / chunks:8.1f}  rest {t[w, 3] / chunks:8.1f}   total {t[w].sum():10.0f}")
