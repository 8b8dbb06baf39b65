"""Condenses a rocprofv3 kernel_stats CSV: short kernel names, calls, total / average time.  usage: kstats.py <csv> [steps]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
out = []
for r in rows:
    name = r["Name"]
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)
    short = re.sub(r"void ", "", short)[:70]
    out.append((float(r["TotalDurationNs"]), int(r["Calls"]), float(r["AverageNs"]), short))
out.sort(reverse=True)
tot = sum(o[0] for o in out)
print(f"total kernel time {tot / 1e6:.2f} ms  ({tot / 1e6 / steps:.2f} ms per step over {steps:g} steps)")
for t, c, a, n in out[:40]:
    print(f"{t / 1e6 / steps:9.3f} ms/step {c / steps:8.1f} calls/step {a / 1e3:9.1f} us avg  {n}")
