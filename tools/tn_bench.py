"""Microbenchmark of the weight-gradient GEMM (gcpnet_tn_gemm): median launch time over HIP events, TFLOP/s and operand GB/s.
usage: tn_bench.py [rows M N]...   (default: the ResGCP weight gradients of configs[1] and configs[4])"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gcpnet_amd import ops  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(159913, 128, 144), (999995, 256, 284), (999995, 128, 144)]
forms = [("bf16x3", None), ("fp32", "1")]  # (GCPNET_TN_PIPE=0 in the environment: the kernels of rounds 3 - 4)
for rows, M, N in [(r, m, n) for (r, m, n) in shapes for _ in forms]:
    form = forms[0] if not hasattr(sys, "_tn_i") or sys._tn_i % 2 == 0 else forms[1]
    sys._tn_i = getattr(sys, "_tn_i", 0) + 1
    os.environ.pop("GCPNET_TN_FP32", None)
    if form[1]:
        os.environ["GCPNET_TN_FP32"] = form[1]
    a = torch.randn(rows, M, device="cuda")
    b = torch.randn(rows, N, device="cuda")
    for _ in range(3):
        ops._tn_weight_grad(a, b)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    torch.cuda.synchronize()
    for s, e in ev:
        s.record()
        ops._tn_weight_grad(a, b)
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    t = ts[len(ts) // 2] * 1e-3
    print(f"{form[0]:6s} rows {rows} M {M} N {N}: {t * 1e3:8.3f} ms  {2.0 * rows * M * N / t / 1e12:6.1f} TFLOP/s  operands {4.0 * rows * (M + N) / t / 1e9:7.0f} GB/s", flush=True)
