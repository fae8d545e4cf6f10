"""A plain C99 caller (tests/c_abi/dropin_test.c) written against include/goldilocks_amd.h the way the
reference's users write code: it must compile with -std=c99 -pedantic and link against the in-tree
library (CPU check), and pass on the GPU (gpu check)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "dropin_test.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "dropin_test.bin")
LIBDIR = os.path.join(ROOT, "libgoldilocks_amd")


def build():
    import libgoldilocks_amd as ga
    if not os.path.exists(ga.LIB_PATH):
        import __graft_entry__ as g
        g.build_lib()
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-o", EXE, SRC, "-L", LIBDIR, "-lgoldilocks_amd", "-Wl,-rpath," + LIBDIR])
    return EXE


def test_c_program_compiles_and_links():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_c_program_runs_on_the_gpu():
    exe = build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dropin_test ok" in r.stdout, (r.stdout, r.stderr)
