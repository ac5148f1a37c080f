"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into bytes per launch per kernel family.
Corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so wide coalesced reads are doubled; WRITE_SIZE is exact for 16-byte stores."""
import csv, glob, json, re, sys, collections
root = sys.argv[1]
MODE = {0: "fwd", 1: "dgrad", 2: "stem", 3: "fwd3x3patch", 4: "dgrad3x3patch"}
PRO = {0: "none", 1: "bn_relu", 2: "bn_bwd"}
EPI = {0: "plain", 1: "stats", 2: "bnbwd_stats", 3: "fc", 4: "bnbwd_stats_maskout"}
def label(name):
    # k_igemm<T, BM, BN, WM, WN, MODE, PRO, EPI, ADD, KC, PD, NS, PERSIST, SPEC>: the class label of frx/ops.py: igemm_class
    m = re.search(r"k_igemmI(DF16b|f)((?:L[ib]\d+E)+)", name)
    if m:
        v = [int(x) for x in re.findall(r"L[ib](\d+)E", m.group(2))]
        bm, bn, wm, wn, mode, pro, epi, add, kc, pd, ns, persist, spec = v[:13]
        stage = "patch" if mode in (3, 4) else (f"dma{ns}" if ns else "ring")
        return (f"k_igemm<{'bf16' if m.group(1)=='DF16b' else 'f32'},{bm}x{bn}x{wm * wn}w,kc{kc},{MODE[mode]},pro={PRO[pro]},epi={EPI[epi]}"
                f"{'+add' if add else ''},{stage}{',persist' if persist else ''}{',stagewaves' if spec else ''}>")
    # the row-resident / streamed pointwise kernels (csrc/pw_rows.hip, pw_stream.hip): mangled or demangled
    m = re.search(r"k_pw_rows_dgradILi\d+ELb([01])E|k_pw_rows_dgrad<\d+, (true|false)", name)
    if m:
        add = (m.group(1) == "1") if m.group(1) is not None else (m.group(2) == "true")
        return f"k_pw_rows<bf16,64x128x8w,dgrad,pro=bn_bwd,epi=bnbwd_stats_maskout{'+add' if add else ''},rows>"
    if "k_pw_rows_fwd" in name: return "k_pw_rows<bf16,64x128x4w,fwd,pro=bn_relu,epi=stats,rows>"
    m = re.search(r"k_pw_streamILi\d+ELi(\d+)ELb([01])E|k_pw_stream<\d+, (\d+), (true|false)", name)
    if m:
        n = m.group(1) or m.group(3)
        dg = (m.group(2) == "1") if m.group(2) is not None else (m.group(4) == "true")
        return (f"k_pw_stream<bf16,16x{n}x8w,dgrad,pro=bn_bwd,epi=bnbwd_stats,stream>" if dg else
                f"k_pw_stream<bf16,16x{n}x8w,fwd,pro=none,epi=stats,stream>")
    if "k_wgrad_grouped" in name:          # (both instantiations, mangled or demangled: frx/ops.py labels them as one class)
        return "k_wgrad_grouped<f32>" if re.search(r"k_wgrad_groupedIf|k_wgrad_grouped<float", name) else "k_wgrad_grouped<bf16>"
    m = re.search(r"k_wgradI(DF16b|f)Li(\d+)E", name)
    if m: return f"k_wgrad<{'bf16' if m.group(1)=='DF16b' else 'f32'},{m.group(2)}>"
    return re.sub(r"\(.*", "", name)[:60]
tot = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
cnt = collections.defaultdict(int)
for c in tot:
    for f in glob.glob(f"{root}/{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = label(r["Kernel_Name"])
                tot[c][k] += float(r["Counter_Value"])
                if c == "FETCH_SIZE": cnt[k] += 1
out = {}
for k, n in cnt.items():
    rd = 2.0 * tot["FETCH_SIZE"][k] * 1024 / n          # gfx950 half-count correction
    wr = tot["WRITE_SIZE"][k] * 1024 / n
    out[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
              "hbm_bytes_per_launch": round(rd + wr)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])), indent=1))
