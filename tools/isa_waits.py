"""Scans hipcc's assembly (`hipcc -S --cuda-device-only ...`) of one kernel for loads that are waited for right after they
are issued (`s_waitcnt vmcnt(0)` within a few instructions and no other load in between): the signature of a guarded load
(`cond ? p[i] : 0`) or of a spill next to an in-flight load, each of which costs a full memory round trip.
usage: isa_waits.py <file.s> <kernel-name-substring> [window]"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
LOAD = ("global_load", "buffer_load", "scratch_load", "flat_load")
hits = []
for i, l in enumerate(out):
    if l.startswith(LOAD):
        for j in range(i + 1, min(i + win, len(out))):
            if out[j].startswith(LOAD):
                break
            if "s_waitcnt" in out[j] and "vmcnt(0)" in out[j]:
                hits.append(i)
                break
print(f"{len(out)} instructions, {sum(l.startswith('v_mfma') for l in out)} MFMAs, {len(hits)} loads waited for on the spot")
print(Counter(out[i].split()[0] for i in hits))
print("instruction indices:", hits)
if len(sys.argv) > 4:  # context dump around the given instruction indices
    for c in (int(x) for x in sys.argv[4].split(",")):
        print(f"---- around {c}")
        for l in out[max(0, c - 14):c + 10]:
            print("   ", l[:100])
