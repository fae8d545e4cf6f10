#!/usr/bin/env python3
"""Per-kernel register / scratch figures straight from the code-object notes of the built library.

    python tools/kernel_resources.py [--json out.json] [lib.so]

Unbundles the gfx950 code object from the .so's .hip_fatbin section (clang-offload-bundler) and reads
the AMDGPU metadata note (llvm-readelf --notes): .vgpr_count, .agpr_count, .sgpr_count,
.vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane),
.group_segment_fixed_size (LDS bytes per block).  No GPU needed."""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(lib):
    import yaml
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"     # one bundle per translation unit, concatenated
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            part = os.path.join(tmp, "bundle%d" % i)
            open(part, "wb").write(blob[starts[i]:starts[i + 1]])
            co = os.path.join(tmp, "gfx950_%d.co" % i)
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                                   "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
            m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", notes, re.S | re.M)
            if not m:
                continue
            meta = yaml.safe_load(m.group(1))
            for k in meta.get("amdhsa.kernels", []):
                out[k[".name"]] = {key[1:]: k[key] for key in (
                    ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
                    ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size") if key in k}
    return out


def main():
    args = sys.argv[1:]
    js = None
    if args and args[0] == "--json":
        js, args = args[1], args[2:]
    lib = args[0] if args else os.path.join(ROOT, "libgoldilocks_amd", "libgoldilocks_amd.so")
    res = resources(lib)
    cols = ["vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size",
            "group_segment_fixed_size"]
    print("%-34s %5s %5s %5s %6s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratchB", "ldsB"))
    for k in sorted(res):
        print("%-34s %5d %5d %5d %6d %8d %7d" % tuple([k] + [res[k].get(c, -1) for c in cols]))
    if js:
        json.dump(res, open(js, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
