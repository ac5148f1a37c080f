"""Compare scripts/layer_times.py outputs per (op, conv shape): mean over the occurrences of the shape in one step.
Usage: python scripts/lt_compare.py base.txt other.txt [more.txt ...]"""
import sys
from collections import defaultdict
def rows(f):
    out = []
    for l in open(f):
        p = l.split()
        if len(p) >= 10 and p[0].startswith("conv"):      # op Ci Co k s Hi us TF/s GB/s class
            out.append((p[0].replace("conv_fwd_tot", "conv_fwd"), p[9], tuple(p[1:6]), float(p[6])))
    return out
fs = sys.argv[1:]
R = [rows(f) for f in fs]
agg, lab = defaultdict(lambda: [[] for _ in fs]), {}
for k, rs in enumerate(R):
    for r in rs:
        agg[(r[0], r[2])][k].append(r[3])
        if k == 0:
            lab[(r[0], r[2])] = r[1]
tot = [0.0] * len(fs)
for key, v in agg.items():
    m = [sum(x) / max(len(x), 1) for x in v]
    n = len(v[0])
    for k in range(len(fs)):
        tot[k] += m[k] * n
    d = (min(m[1:]) - m[0]) * n if len(m) > 1 else 0.0
    flag = f"   {d:+6.0f} us/step" if abs(d) > 0.03 * m[0] * n else ""
    print(f"{key[0]:14s} {' '.join(key[1]):22s} x{n} {lab[key]:22s} " + " | ".join(f"{x:6.1f}" for x in m) + flag)
print("sum us/step:", [round(t) for t in tot])
