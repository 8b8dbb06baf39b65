"""Kernel resource usage of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
VGPRs / AGPRs / SGPR + VGPR spills / scratch bytes / occupancy.   usage: python tools/kres.py gcpnet_amd/csrc/<file>.hip [filter] [-D...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.split("\n"):
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
    if "error:" in line:
        print(line)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name).split("(")[0]
    if flt and flt not in name:
        continue
    print(f"{name:60s} VGPR {v.get('VGPRs'):>4s} AGPR {v.get('AGPRs'):>3s} spillV {v.get('VGPRs Spill'):>3s} spillS {v.get('SGPRs Spill'):>3s} "
          f"scratch {v.get('ScratchSize'):>4s} occ {v.get('Occupancy')} sgpr {v.get('TotalSGPRs')}")
