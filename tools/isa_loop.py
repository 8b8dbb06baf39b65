"""Instruction-category string of the main loop (between the last two s_barrier) of one kernel in hipcc's assembly, plus its vmcnt waits.
usage: isa_loop.py <file.s> <kernel-name-substring> [index of the barrier that opens the loop]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
out, on = [], False
for l in open(path):
    if re.match(r"^_Z\w*:", l):
        on = key in l
    if on:
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            out.append(t)
        if "s_endpgm" in l:
            on = False
bars = [i for i, l in enumerate(out) if l.startswith("s_barrier")]
print(len(out), "instructions; barriers at", bars)
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(bars) - 2
seg = out[bars[k]:bars[k + 1] + 1]
print(collections.Counter(x.split()[0] for x in seg).most_common(16))
cat = []
for l in seg:
    o = l.split()[0]
    cat.append("M" if o.startswith("v_mfma") else "L" if o.startswith(("global_load", "buffer_load")) else "R" if o.startswith("ds_read") else
               "W" if o.startswith("ds_write") else "w" if o.startswith("s_waitcnt") else "b" if "branch" in o else "v" if o.startswith("v_") else "s")
print("".join(cat))
print([l.split(None, 1)[1] for l in seg if l.startswith("s_waitcnt") and "vmcnt" in l])
