"""Table of what the element-wise comparisons of a GPU test run achieved, from the report tests/helpers.close() writes when
GCPNET_PARITY_REPORT=<file> is set:   GCPNET_PARITY_REPORT=/tmp/p.tsv python -m pytest tests -m gpu -q; python tools/parity_table.py /tmp/p.tsv
Per test: comparisons, the largest error in units of the north_star tolerance (1e-5 absolute + 1e-5 relative, element-wise) and the
loosest tolerance the test allows itself."""
import collections
import sys

rows = collections.defaultdict(list)
for line in open(sys.argv[1]):
    t, shape, err, scale, atol, rtol, unit = line.rstrip("\n").split("\t")
    rows[t].append((float(unit), float(err), float(scale), float(atol), float(rtol)))
print(f"{'test':110s} {'n':>4s} {'max err / (1e-5 + 1e-5|ref|)':>30s} {'max abs err':>12s} {'loosest atol / rtol':>20s}")
over = 0
for t in sorted(rows):
    v = rows[t]
    u = max(x[0] for x in v)
    over += u > 1.0
    print(f"{t[:110]:110s} {len(v):4d} {u:30.3f} {max(x[1] for x in v):12.3e} {max(x[3] for x in v):9g} / {max(x[4] for x in v):g}")
print(f"{len(rows)} tests with element-wise comparisons; {over} of them hold a comparison looser than 1e-5 + 1e-5 |ref| that needed it")
