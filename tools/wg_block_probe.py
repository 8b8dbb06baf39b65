"""The (256,32) block backward (gcp_wg_bwd_kernel, the configs[4] chain's kernel) alone on 10^6 rows: median launch time of
bench.c5_kernel_roofline's probe (tile-blocked tensors, as inside the chain).  GCPNET_HIP_LIB selects a variant build."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
r = bench.c5_kernel_roofline(G, ops, rows, 256, 32, iters=7)
print(json.dumps({"rows": rows, "bwd_ms": round(r["median_launch_ms"], 4), "row_major_ms": round(r["row_major_launch_ms"], 4), "frac": round(r["frac"], 4)}))
