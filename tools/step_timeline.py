"""Timeline of the last bench step in a rocprofv3 kernel_trace CSV: every launch with its start offset, duration and queue, the busy /
idle time of the busiest queue (the caller's stream) and the launches that run beside it.
usage: step_timeline.py <kernel_trace.csv> <steps incl. warmup> [min_us to print] [anchor:K]
anchor:K = instead of cutting the trace into equal parts, print the launches between the 2K-th and the K-th launch from the end whose
kernel name contains `anchor` (K launches of it per step: one step, rotated to start at that kernel) -- robust against set-up launches."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // steps
last = rows[-per:]
if len(sys.argv) > 4:
    name, k = sys.argv[4].rsplit(":", 1)
    hits = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
    last = rows[hits[-2 * int(k)]:hits[-int(k)]]
    per = len(last)
t0 = int(last[0]["Start_Timestamp"])
byq = collections.defaultdict(float)
for r in last:
    byq[r["Queue_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
main = max(byq, key=byq.get)
end_prev, idle, gaps = None, 0.0, []
for r in last:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:70]
    q = "M" if r["Queue_Id"] == main else "s"
    gap = ""
    if q == "M":
        if end_prev is not None and s > end_prev:
            idle += s - end_prev
            gap = f"  (+{s - end_prev:.1f} idle)"
            gaps.append(s - end_prev)
        end_prev = max(end_prev or 0.0, e)
    if e - s >= min_us:
        print(f"{s:10.1f} us {e - s:8.1f} us {q} grid {r['Grid_Size_X']:>8} {n}{gap}")
span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
print(f"step span {span / 1e3:.2f} ms, {per} launches; kernel time per queue (ms): { {('main' if q == main else q): round(t / 1e3, 2) for q, t in byq.items()} }; "
      f"main queue idle between its launches: {idle / 1e3:.2f} ms in {len(gaps)} gaps (largest {max(gaps) if gaps else 0:.0f} us)")
