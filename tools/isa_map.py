"""One kernel of a hipcc assembly file as a string of events -- where the matrix instructions, memory accesses, branches and full
memory drains sit relative to each other.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S gcpnet_amd/csrc/gcp2_chain_bwd.hip -o /tmp/cb.s
  python tools/isa_map.py /tmp/cb.s gcp2_chain_bwd_kernelILi4ELi2ELb1ELi4ELb1

  B v_mfma_f32_32x32x16_bf16   F v_mfma_f32_32x32x2_f32   s v_mfma_f32_16x16x4_f32
  r global load   w global store   l LDS read   d LDS write   S / L scratch (spill) store / load
  | branch        ! s_waitcnt vmcnt(0)
Runs of four or more equal events are written x{n}.  What to look for: `r!r!r!` (loads waited for one by one:
spilled destinations, or a use right behind each request), `S` / `L` inside the block loop, long runs of `|` around stores (the
general tile helpers' fallback paths)."""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:\s", l)]
if not starts:
    sys.exit(f"no kernel matching {pat}")
start = starts[0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
KINDS = [("v_mfma_f32_32x32x16", "B"), ("v_mfma_f32_32x32x2", "F"), ("v_mfma_f32_16x16x4", "s"), ("s_cbranch", "|"), ("s_branch", "|"),
         ("global_store", "w"), ("global_load", "r"), ("ds_write", "d"), ("ds_store", "d"), ("ds_read", "l"), ("ds_load", "l"),
         ("s_waitcnt vmcnt(0)", "!"), ("scratch_store", "S"), ("scratch_load", "L")]
ev = []
for l in lines[start:end]:
    t = l.strip()
    for prefix, k in KINDS:
        if t.startswith(prefix):
            ev.append(k)
            break
out = re.sub(r"(.)\1{3,}", lambda m: f"{m.group(1)}{{{len(m.group(0))}}}", "".join(ev))
print(lines[start].split(":")[0])
print(f"{end - start} lines")
print(out)
print({k: ev.count(k) for k in "BFs|!SLrwld"})
