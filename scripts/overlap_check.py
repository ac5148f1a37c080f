"""Do the grouped weight-gradient launches overlap other kernels in time?  Reads a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
wg = [k for k in ks if "k_wgrad_grouped" in k[2]]
print("kernels", len(ks), "wgrad launches", len(wg))
tot = ov = 0
for s, e, n in wg[-9:]:
    o = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2 in ks if n2 is not n and e2 > s and s2 < e and "k_wgrad_grouped" not in n2)
    tot += e - s; ov += o
    print(f"wgrad {(e - s) / 1e3:8.1f} us, other kernels running during it: {o / 1e3:8.1f} us")
print(f"overlap fraction {ov / max(tot, 1):.2f}")
