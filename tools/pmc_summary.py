"""Aggregates a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(path)):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"^void ", "", k)[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:8]:
    n = len(cnt[k])
    print(f"{k:44s} n={n:4d} " + "  ".join(f"{c}={v / n:.3g}" for c, v in sorted(acc[k].items())))
