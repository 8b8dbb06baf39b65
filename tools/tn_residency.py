"""Which workgroups of a pipelined weight-gradient GEMM launch ran WHEN and WHERE: start / end s_memtime, CU (HW_ID) and XCD of every
workgroup through the library's phase-timing hook.  Prints the launch's span, the per-workgroup duration, and how many workgroups
were resident on a CU at the same time.   usage: tn_residency.py rows M N [problems per launch]"""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gcpnet_amd import _lib, ops  # noqa: E402

rows, M, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (999995, 128, 144)
nprob = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib = _lib.load()
items = [(torch.randn(rows, M, device="cuda"), torch.randn(rows, N, device="cuda"), torch.empty(M, N, device="cuda")) for _ in range(nprob)]
for _ in range(2):
    ops._tn_weight_grads_into(items)
cap = 8192
buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), cap)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops._tn_weight_grads_into(items)
e1.record()
torch.cuda.synchronize()
wall_ms = e0.elapsed_time(e1)
lib.gcpnet_debug_set_phase_timing(None, 0)
t = buf.view(cap, 8).cpu()
t = t[t[:, 0] > 0]
t0 = int(t[:, 0].min())
start, end = (t[:, 0] - t0).double(), (t[:, 1] - t0).double()
hw, xcc = t[:, 2], t[:, 3] & 0xf
cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7) + 128 * ((hw >> 12) & 1) + 256 * xcc  # cu_id, se_id, sh_id, xcd -> one key per CU
print(f"{len(t)} workgroups; duration mean {float((end - start).mean()):.0f} min {float((end - start).min()):.0f} max {float((end - start).max()):.0f}")
spans = [float((t[xcc == k, 1].max() - t[xcc == k, 0].min())) for k in range(8) if (xcc == k).any()]
print(f"wall {wall_ms:.3f} ms (incl. the reduction kernel); span inside an XCD (its own counter): {min(spans):.0f} - {max(spans):.0f} ticks -> {max(spans) / wall_ms / 1e6:.2f} G ticks/s")
print(f"distinct CUs {len(set(cu.tolist()))}; workgroups per XCD {sorted(collections.Counter(xcc.tolist()).items())}")
# concurrency per CU: for each workgroup, how many others on the same CU overlap its midpoint
by = collections.defaultdict(list)
for i, c in enumerate(cu.tolist()):
    by[c].append((float(start[i]), float(end[i])))
conc = collections.Counter()
for c, iv in by.items():
    for s, e in iv:
        mid = 0.5 * (s + e)
        conc[sum(1 for s2, e2 in iv if s2 <= mid < e2)] += 1
print("workgroups resident on the same CU at a workgroup's midpoint:", sorted(conc.items()))
print("start times (first 12 sorted):", [int(x) for x in sorted(start.tolist())[:12]], "... last start", int(start.max()))
