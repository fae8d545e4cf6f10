"""CPU tests of the drop-in boundary: the C-ABI shared library loads without a GPU and exports every
symbol include/goldilocks_amd.h declares (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "goldilocks_amd.h")


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    funcs = re.findall(r"GOLDILOCKS_AMD_API\s+[\w\s\*]+?\b(goldilocks_\w+)\s*\(", text)
    data = re.findall(r"GOLDILOCKS_AMD_API\s+extern\s+const\s+[\w\s\*]+?\b(goldilocks_\w+)\s*(?:\[[^\]]*\])?\s*;", text)
    return sorted(set(funcs)), sorted(set(data))


@pytest.fixture(scope="module")
def L():
    import libgoldilocks_amd as ga
    if not os.path.exists(ga.LIB_PATH):
        import __graft_entry__ as g
        g.build_lib()
    return ga.lib()


def test_header_and_binding_agree():
    import libgoldilocks_amd as ga
    funcs, data = declared_symbols()
    assert len(funcs) >= 40 and len(data) == 8
    assert sorted(ga.FUNCTIONS) == funcs
    assert sorted(ga.DATA_SYMBOLS) == data


def test_library_exports_every_declared_symbol(L):
    funcs, data = declared_symbols()
    for name in funcs:
        assert getattr(L, name) is not None
    for name in data:
        C.c_char.in_dll(L, name)


def test_exported_constants_match_the_reference(L):
    import json
    import numpy as np
    import libgoldilocks_amd as ga
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "f5_constants.json")))
    assert C.c_size_t.in_dll(L, "goldilocks_448_sizeof_precomputed_s").value == k["sizeof_precomputed_s"]
    # the reference exports 16 (generic 64-bit build, the fixture) or 32 (AVX2): we export the stricter one
    align = C.c_size_t.in_dll(L, "goldilocks_448_alignof_precomputed_s").value
    assert align == 32 and align % k["alignof_precomputed_s"] == 0
    assert [int(x) for x in ga.point_base()] == k["point_base_limbs"]
    import hashlib
    assert hashlib.sha256(ga.precomputed_base().tobytes()).hexdigest() == k["precomputed_base_sha256"]
    assert list(ga.point_identity()) == [0] * 8 + [1] + [0] * 7 + [1] + [0] * 7 + [0] * 8
    one = (C.c_uint64 * 7).in_dll(L, "goldilocks_448_scalar_one")
    assert list(one) == [1, 0, 0, 0, 0, 0, 0]


def test_no_cpu_fallback_without_gpu(L):
    """Without a device the library reports an error (batch API) -- it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    import libgoldilocks_amd as ga
    with pytest.raises(ga.GoldilocksAmdError):
        ga.point_scalarmul_batch(np.zeros((1, 32), np.uint64), np.zeros((1, 7), np.uint64))
    assert b"hip" in L.goldilocks_amd_last_error().lower()


def test_code_object_is_gfx950_only(L):
    import libgoldilocks_amd as ga
    out = subprocess.run(["strings", "-a", ga.LIB_PATH], capture_output=True, text=True).stdout
    archs = set(re.findall(r"amdgcn-amd-amdhsa--(gfx\w+)", out))
    assert archs == {"gfx950"}, archs


def test_product_sources_do_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "libgoldilocks_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gold_oracle" not in text and "liboracle" not in text and "hostsim" not in text.replace(
                    "tests/hostsim", ""), f


def test_library_exports_nothing_but_the_declared_abi(L):
    """No host-side arithmetic (the checker build's branches of the lane headers), no helper and no
    oracle symbol leaks out of the shared library: its dynamic symbol table is exactly the header."""
    import libgoldilocks_amd as ga
    out = subprocess.run(["nm", "-D", "--defined-only", ga.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split()[1:2] and l.split()[1] in "TDBRVW"}
    declared = set(ga.FUNCTIONS) | set(ga.DATA_SYMBOLS)
    assert exported == declared, sorted(exported ^ declared)      # libgoldilocks_amd/csrc/exports.map
    assert not any("orc_" in s or "hs_" in s or "oracle" in s for s in exported)


def test_void_drop_in_functions_abort_loudly_without_gpu(L):
    """The reference's void functions cannot report errors; without a device ours abort with a
    message instead of returning garbage or computing on the CPU."""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    code = ("import numpy as np, libgoldilocks_amd as ga\n"
            "ga.point_scalarmul(np.zeros(32, np.uint64), np.zeros(7, np.uint64))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == -6, r.returncode                     # SIGABRT
    assert "no CPU path" in r.stderr and "goldilocks_448_point_scalarmul" in r.stderr


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name != "cpu_baseline_leg":
            seg = ast.get_source_segment(src, node)
            code_lines = [l.split("#")[0] for l in seg.splitlines()]
            code = "\n".join(l for l in code_lines if not l.strip().startswith(('"', "'")))
            assert "_libs" not in code and "oracle(" not in code and "oracle_" not in code, node.name
