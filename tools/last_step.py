"""Per-launch durations of the last bench step in a rocprofv3 kernel_trace CSV.  usage: last_step.py <csv> <steps incl. warmup> [min_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // steps
tot = 0.0
for r in rows[-per:]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d >= min_us:
        print(f"{d:9.1f} us  grid {r['Grid_Size_X']:>7}  {n}")
print(f"sum of kernel time in the step: {tot / 1e3:.2f} ms over {per} launches")
