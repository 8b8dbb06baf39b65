"""A kernel's assembly as a compact event list WITH its waits: for the region between two s_memtime stamps (or the whole kernel), one
token per memory / matrix instruction and every s_waitcnt with its counters, plus the VALU / SALU instruction counts in between.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S gcpnet_amd/csrc/gcp2_chain_bwd.hip -o /tmp/cb.s
  python tools/isa_events.py /tmp/cb.s gcp2_chain_bwd_kernelILi4ELi2ELb1ELi4ELb1ELb0

  B bf16 MFMA  F fp32 32x32 MFMA  s fp32 16x16 MFMA  r/w global load/store (x4 = dwordx4)  l/d LDS read/write  T s_memtime
  [v3 g0] = s_waitcnt vmcnt(3) lgkmcnt(0);  a number between events = that many VALU+SALU instructions;  | branch;  --- basic-block label"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:\s", l)]
if not starts:
    sys.exit(f"no kernel matching {pat}")
start = starts[0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
out, n_alu = [], 0


def flush():
    global n_alu
    if n_alu:
        out.append(str(n_alu))
        n_alu = 0


for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            flush(); out.append("\n---" + t.split(":")[0])
        continue
    op = t.split()[0]
    tok = None
    if op.startswith("v_mfma_f32_32x32x16"): tok = "B"
    elif op.startswith("v_mfma_f32_32x32x2"): tok = "F"
    elif op.startswith("v_mfma_f32_16x16x4"): tok = "s"
    elif op.startswith("global_load"): tok = "r" + ("4" if "dwordx4" in op else ("2" if "dwordx2" in op else ""))
    elif op.startswith("global_store"): tok = "w" + ("4" if "dwordx4" in op else ("2" if "dwordx2" in op else ""))
    elif op.startswith("ds_read") or op.startswith("ds_load"): tok = "l" + ("4" if "b128" in op else ("2" if "b64" in op or "read2" in op else ""))
    elif op.startswith("ds_write") or op.startswith("ds_store"): tok = "d" + ("4" if "b128" in op else ("2" if "b64" in op or "write2" in op else ""))
    elif op.startswith("scratch_"): tok = "S" if "store" in op else "L"
    elif op == "s_memtime": tok = "T"
    elif op.startswith("s_cbranch") or op == "s_branch": tok = "|"
    elif op == "s_waitcnt":
        m = re.findall(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", t)
        tok = "[" + " ".join(f"{k[0]}{v}" for k, v in m) + "]"
    elif op.startswith("s_load"): tok = "k"
    if tok is None:
        n_alu += 1
    else:
        flush(); out.append(tok)
print(lines[start].split(":")[0], f"{end - start} lines")
print(" ".join(out))
