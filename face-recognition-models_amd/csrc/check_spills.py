#!/usr/bin/env python3
"""Build-time check: the patch-mode k_igemm kernels (MODE_FWD3 / MODE_DGRAD3) and k_pw_rows_dgrad's loader waves count their own
`s_waitcnt vmcnt(N)` over inline-asm LDS-DMA that hipcc does not model.  A compiler-inserted scratch spill is a
vector-memory operation too: it would shift those counts.  So every such kernel must have no scratch at all:
private_segment_fixed_size == 0 and vgpr_spill_count == 0 in the code object's metadata.  Fails the build otherwise.

usage: check_spills.py <object.o> [...]   (host objects with an embedded gfx950 code object)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def kernels(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            cur = {"name": m.group(2)}
            out.append(cur)
        else:
            cur[m.group(1)] = int(m.group(2))
    return out


def counts_vmcnt(name):
    """kernels that wait on hand-counted vmcnt: the patch-mode k_igemm and the loader waves of k_pw_rows_dgrad"""
    return is_patch_mode(name) or "k_pw_rows_dgrad" in name


def is_patch_mode(name):
    # k_igemm<T, BM, BN, WM, WN, MODE, ...>: MODE is the fifth integer template argument (3 = MODE_FWD3, 4 = MODE_DGRAD3)
    m = re.match(r"_ZN3frx7k_igemmI(?:DF16b|f)((?:Li\d+E)+)", name)
    if not m:
        return False
    ints = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
    return len(ints) >= 5 and ints[4] in (3, 4)


def main(objs):
    bad, seen = [], 0
    for obj in objs:
        for k in kernels(obj):
            if not counts_vmcnt(k["name"]):
                continue
            seen += 1
            if k.get("private_segment_fixed_size", 0) != 0 or k.get("vgpr_spill_count", 0) != 0:
                bad.append(k)
    for k in bad:
        print(f"check_spills: {k['name']}: scratch {k.get('private_segment_fixed_size')} B, vgpr spills {k.get('vgpr_spill_count')}",
              file=sys.stderr)
    if bad or not seen:
        print(f"check_spills: FAILED ({len(bad)} of {seen} kernels with hand-counted vmcnt use scratch)" if seen else
              "check_spills: no patch-mode kernel found (name pattern changed?)", file=sys.stderr)
        return 1
    print(f"check_spills: {seen} kernels with hand-counted vmcnt, none uses scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
