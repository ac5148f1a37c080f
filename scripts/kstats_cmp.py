"""Compare two rocprofv3 kernel_stats.csv files (scripts/kstats_ab.sh) per kernel: calls and ms per step."""
import csv, re, sys
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 39
def short(n):
    n = re.sub(r'\(.*', '', n).replace('void frx::', '').replace('frx::', '')
    m = re.match(r'_ZN3frx7k_igemmIDF16bLi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d+)ELi\d+ELi(\d)EEE', n)
    if m:
        return "k_igemm<bf16,%sx%s,mode%s,pro%s,epi%s,add%s,kc%s,ns%s>" % m.groups()
    return n[:70]
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1]))
ta = tb = 0
print(f"{'kernel':72s} {'A calls':>8s} {'ms/step':>8s} | {'B calls':>8s} {'ms/step':>8s}   A-B")
for k in keys:
    ca, ma = a.get(k, (0, 0)); cb, mb = b.get(k, (0, 0)); ta += ma; tb += mb
    if max(ma, mb) / steps > 0.008:
        print(f"{short(k):72s} {ca/steps:8.1f} {ma/steps:8.3f} | {cb/steps:8.1f} {mb/steps:8.3f}  {(ma-mb)/steps*1e3:+7.0f} us")
print("total kernel ms/step", round(ta / steps, 3), round(tb / steps, 3))
