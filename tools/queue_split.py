"""Kernel time per HIP queue (stream) and per kernel on the busiest queue, averaged over the steps of a bench run.
usage: queue_split.py <kernel_trace.csv> <steps incl. warmup>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
byq = collections.defaultdict(float)
for r in rows:
    byq[r["Queue_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print({q: round(t / steps, 2) for q, t in byq.items()}, "ms of kernel time per step and queue")
main = max(byq, key=byq.get)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r["Queue_Id"] != main:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:56]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{t / steps:7.3f} ms/step {c / steps:6.1f} calls/step  {n}")
