"""The weight-gradient GEMM as a training step launches it: ONE gcpnet_tn_gemm call with four `scalar_out` problems and four gate
problems of the residual message GCPs (enough workgroups to fill every CU: per-CU throughput decides, not per-workgroup latency).
usage: tn_mix_bench.py [c2|c5|c3]...   prints median ms per call (incl. the reduction), fp32-equivalent TFLOP/s and operand GB/s"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gcpnet_amd import ops  # noqa: E402

MIX = {"c2": (159913, [(128, 144), (16, 128)]), "c5": (999995, [(256, 276), (32, 256)]), "c3": (199746, [(100, 128), (16, 100)]),
       "c2n": (10000, [(512, 144), (128, 532)]), "c5n": (100000, [(1024, 276), (256, 896)]),
       "c2s": (159913, [(128, 144)]), "c2g": (159913, [(16, 128)]), "c5s": (999995, [(256, 276)]), "c5g": (999995, [(32, 256)]),
       "c1": (2000, [(128, 144), (16, 128)]), "c1n": (500, [(512, 144), (128, 532)]),
       "c4": (38000, [(128, 144), (16, 128)]), "c4n": (2000, [(512, 144), (128, 532)])}
for name in (sys.argv[1:] or ["c2", "c5"]):
    rows, shapes = MIX[name]
    items = []
    for M, N in shapes * (8 // len(shapes)):
        items.append((torch.randn(rows, M, device="cuda"), torch.randn(rows, N, device="cuda"), torch.empty(M, N, device="cuda")))
    for _ in range(3):
        ops._tn_weight_grads_into(items)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    torch.cuda.synchronize()
    for s, e in ev:
        s.record()
        ops._tn_weight_grads_into(items)
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    t = ts[len(ts) // 2] * 1e-3
    fl = sum(2.0 * rows * M * N for M, N in shapes * (8 // len(shapes)))
    by = sum(4.0 * rows * (M + N) for M, N in shapes * (8 // len(shapes)))
    print(f"{name}: {t * 1e3:8.3f} ms  {fl / t / 1e12:6.1f} TFLOP/s  operands {by / t / 1e9:7.0f} GB/s", flush=True)
