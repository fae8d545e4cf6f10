#!/usr/bin/env python3
"""Build experimental variants of the library into variants/ (git-ignored .so files that travel to the
GPU box):   python tools/build_variants.py name1:-DX=1,-DY=2 name2:...     (see tools/probes/variant_sweep.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
for spec in sys.argv[1:]:
    name, _, defs = spec.partition(":")
    defines = [d[2:] for d in defs.split(",") if d.startswith("-D")]
    flags = [d for d in defs.split(",") if d and not d.startswith("-D")]
    g.build_lib(force=True, defines=tuple(defines), out=os.path.join(ROOT, "variants", "libgoldilocks_amd_%s.so" % name),
                extra_flags=tuple(flags))
