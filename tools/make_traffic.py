"""profiles/<tag>_traffic.json from the PMC passes of tools/profile_round.sh (rocprofv3 --pmc, counter_collection.csv files):
HBM-side bytes per launch and MFMA-busy share of the kernels bench.py reports a roofline for.

  python tools/make_traffic.py gpurun_out/<tag> profiles/<tag>_traffic.json

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) x 1024 x 2 (128-byte requests are
tallied at 64 bytes), WRITE_SIZE (KB) x 1024 as reported.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

root, out_path = sys.argv[1], sys.argv[2]


def load(sub):
    """{short kernel name: {grid: {counter: [values per dispatch]}}}"""
    path = None
    for d, _, fs in os.walk(os.path.join(root, sub)):
        for f in fs:
            if f.endswith("counter_collection.csv"):
                path = os.path.join(d, f)
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(lambda: defaultdict(float))))
    if path is None:
        return acc
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k)
        acc[k][int(r.get("Grid_Size", 0))][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return acc


def mean(d):
    return sum(d.values()) / max(len(d), 1)


def pick(acc, prefix, grid_rank=None):
    """Counter means of the kernel whose name starts with `prefix`; grid_rank: only its `grid_rank` largest grids (summed)."""
    rows = [(k, g, c) for k, gs in acc.items() if k.startswith(prefix) for g, c in gs.items()]
    if not rows:
        return {}
    rows.sort(key=lambda t: -max(mean(d) for d in t[2].values()))  # (heaviest launches first: by the counter itself, not by grid size)
    if grid_rank:
        rows = rows[:grid_rank]
        out = defaultdict(float)
        for _, _, c in rows:
            for name, d in c.items():
                out[name] += mean(d)
        return dict(out)
    # (one kernel, several template instantiations or grids: the one with the most dispatches)
    best = max(rows, key=lambda t: sum(len(d) for d in t[2].values()))
    return {name: mean(d) for name, d in best[2].items()}


res = {"source": f"{root}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE, separate passes, "
                 "no tracing, `bench.py --steps 3 --warmup 1 --step-only` (c2) and `--config c5 --steps 2 --warmup 1 --step-only`; mean per dispatch",
       "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes), WRITE_SIZE as reported; KB x 1024",
       "mfma_busy_frac": {"note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}}
targets = [("gcp2_chain_bwd_kernel", "", "gcp2_chain_bwd_kernel", None), ("gcp2_chain_fwd_kernel", "", "gcp2_chain_fwd_kernel<4, true, false, false", None),
           ("tn_pipe_kernel(+reduce)", "", "tn_pipe_kernel<4", 2), ("gcp_wg_bwd_kernel (256,32)", "_c5", "gcp_wg_bwd_kernel<8, 1, 0, true, 2", None)]
for key, suffix, prefix, rank in targets:
    f = pick(load("pmc_FETCH_SIZE" + suffix), prefix, rank).get("FETCH_SIZE")
    w = pick(load("pmc_WRITE_SIZE" + suffix), prefix, rank).get("WRITE_SIZE")
    m = pick(load("pmc_mfma" + suffix), prefix, rank)
    if f is not None and w is not None:
        res[key] = 2 * f * 1024 + w * 1024
        res[key + "_parts"] = {"fetch_x2_bytes": 2 * f * 1024, "write_bytes": w * 1024,
                               "note": f"2 x {f * 1024 / 1e9:.3f} + {w * 1024 / 1e9:.3f} GB per launch" + (" (the two launches of a layer's chain)" if rank else "")}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
        res["mfma_busy_frac"][key] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
