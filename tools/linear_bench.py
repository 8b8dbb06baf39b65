"""Plain Linear on node rows: ops.wg_linear (workgroup kernel) against torch.matmul (rocBLAS), median of 15 launches.
usage: linear_bench.py [rows in out]..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gcpnet_amd import ops  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(100000, 256, 256), (100000, 256, 512), (10000, 128, 128), (100000, 896, 256)]


def med(fn):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    torch.cuda.synchronize()
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e-3


for rows, din, dout in shapes:
    x = torch.randn(rows, din, device="cuda")
    W = torch.randn(dout, din, device="cuda") * 0.05
    fl = 2.0 * rows * din * dout
    out = ops.wg_linear(x, W, dout, din)
    if out is None:
        print(f"rows {rows} {din}->{dout}: wg_linear refuses the shape")
        t_wg = float("nan")
    else:
        t_wg = med(lambda: ops.wg_linear(x, W, dout, din))
    t_bl = med(lambda: torch.matmul(x, W.t()))
    print(f"rows {rows} {din}->{dout}: wg_linear {t_wg * 1e3:7.3f} ms {fl / t_wg / 1e12:6.1f} TFLOP/s   rocBLAS {t_bl * 1e3:7.3f} ms {fl / t_bl / 1e12:6.1f} TFLOP/s")
