"""The host checker build of the lane arithmetic (tests/hostsim) once more under AddressSanitizer and
UndefinedBehaviorSanitizer: out-of-bounds table / scalar-word / staging accesses, shifts by the word
size, signed overflow in the carry chains and the lattice reduction would abort the child process.
(GPU sanitizers are not available on the MI355X pool: this is the sanitizer coverage the device code gets,
through the host compile of the same headers.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HS_DIR = os.path.join(HERE, "hostsim")
SELECT = "ladders_and_codecs or verify or half_size or four_bit_window or fixed_base_window or big_comb"


def test_lane_arithmetic_under_asan_and_ubsan():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", HS_DIR, "sanitize"])
    env = dict(os.environ, LD_PRELOAD=asan, GOLDILOCKS_HOSTSIM_LIB=os.path.join(HS_DIR, "libhostsim_san.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_hostsim.py"), "-x", "-q", "-k", SELECT,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
