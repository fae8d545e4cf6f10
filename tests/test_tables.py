"""CPU tests: the generated constant tables (tools/gen_tables.py, independent big-integer math)
against values captured from the reference build and against the oracle's own precompute."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_header_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tables.py"), "--check"])
    assert r.returncode == 0


def test_generator_matches_reference_constants(O):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_tables
    base, comb = gen_tables.generate()
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "f5_constants.json")))
    x, y = base
    limbs = gen_tables.limbs56(x) + gen_tables.limbs56(y) + gen_tables.limbs56(1) + gen_tables.limbs56(x * y)
    assert limbs == k["point_base_limbs"]
    flat = b"".join(int(l).to_bytes(8, "little") for e in comb for f in e for l in gen_tables.limbs56(f))
    assert hashlib.sha256(flat).hexdigest() == k["precomputed_base_sha256"]
    assert flat == bytes(O.orc_precomputed_base().contents)
