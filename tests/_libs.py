"""ctypes bindings used by the tests: the CPU oracle (oracle/liboracle.so) and, when it has
been built in this container, the REAL reference (oracle/_ref/libgoldilocks_ref64.so).

Test infrastructure only.  The product library is bound in libgoldilocks_amd/."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libgoldilocks_ref64.so")
REF_X86_SO = os.path.join(ORACLE_DIR, "_ref", "libgoldilocks_x86_64.so")
REF_X86_V3_SO = os.path.join(ORACLE_DIR, "_ref", "libgoldilocks_x86_64_v3.so")   # -march=x86-64-v3: needs AVX2 + BMI2

P = 2**448 - 2**224 - 1
Q = 2**446 - 0x8335DC163BB124B65129C96FDE933D8D723A70AADC873D6D54A7BB0D


class Gf(C.Structure):
    _fields_ = [("limb", C.c_uint64 * 8)]

    def value(self):
        return sum(int(l) << (56 * i) for i, l in enumerate(self.limb)) % P

    @classmethod
    def from_int(cls, v):
        g = cls()
        for i in range(8):
            g.limb[i] = (v >> (56 * i)) & ((1 << 56) - 1)
        return g


class Point(C.Structure):
    _fields_ = [("x", Gf), ("y", Gf), ("z", Gf), ("t", Gf)]


class Scalar(C.Structure):
    _fields_ = [("limb", C.c_uint64 * 7)]

    def value(self):
        return sum(int(l) << (64 * i) for i, l in enumerate(self.limb))

    @classmethod
    def from_int(cls, v):
        s = cls()
        for i in range(7):
            s.limb[i] = (v >> (64 * i)) & (2**64 - 1)
        return s


class Niels(C.Structure):
    _fields_ = [("a", Gf), ("b", Gf), ("c", Gf)]


class Precomputed(C.Structure):
    _fields_ = [("table", Niels * 80)]


assert C.sizeof(Gf) == 64 and C.sizeof(Point) == 256 and C.sizeof(Scalar) == 56
assert C.sizeof(Precomputed) == 15360


def build_oracle():
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "gold_oracle.c"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_oracle = None


def oracle():
    """Load liboracle.so (building it with gcc if needed) and set prototypes."""
    global _oracle
    if _oracle is not None:
        return _oracle
    L = C.CDLL(build_oracle())
    u8p = C.POINTER(C.c_uint8)
    pg, pp, ps = C.POINTER(Gf), C.POINTER(Point), C.POINTER(Scalar)
    vp = C.c_void_p
    proto = {
        "orc_gf_mul": (None, [pg, pg, pg]),
        "orc_gf_sqr": (None, [pg, pg]),
        "orc_gf_mulw": (None, [pg, pg, C.c_uint32]),
        "orc_gf_add": (None, [pg, pg, pg]),
        "orc_gf_sub": (None, [pg, pg, pg]),
        "orc_gf_strong_reduce": (None, [pg]),
        "orc_gf_isr": (C.c_uint64, [pg, pg]),
        "orc_gf_serialize": (None, [vp, pg]),
        "orc_gf_deserialize": (C.c_uint64, [pg, vp, C.c_uint8]),
        "orc_gf_eq": (C.c_uint64, [pg, pg]),
        "orc_gf_lobit": (C.c_uint64, [pg]),
        "orc_scalar_add": (None, [ps, ps, ps]),
        "orc_scalar_sub": (None, [ps, ps, ps]),
        "orc_scalar_mul": (None, [ps, ps, ps]),
        "orc_scalar_halve": (None, [ps, ps]),
        "orc_scalar_invert": (C.c_int, [ps, ps]),
        "orc_scalar_decode": (C.c_int, [ps, vp]),
        "orc_scalar_decode_long": (None, [ps, vp, C.c_size_t]),
        "orc_scalar_encode": (None, [vp, ps]),
        "orc_point_base": (pp, []),
        "orc_point_identity": (pp, []),
        "orc_precomputed_base": (C.POINTER(Precomputed), []),
        "orc_wnaf_base": (C.POINTER(Niels), []),
        "orc_point_add": (None, [pp, pp, pp]),
        "orc_point_sub": (None, [pp, pp, pp]),
        "orc_point_double": (None, [pp, pp]),
        "orc_point_negate": (None, [pp, pp]),
        "orc_point_debugging_torque": (None, [pp, pp]),
        "orc_point_debugging_pscale": (None, [pp, pp, C.c_void_p]),
        "orc_point_eq": (C.c_int, [pp, pp]),
        "orc_point_valid": (C.c_int, [pp]),
        "orc_point_encode": (None, [vp, pp]),
        "orc_point_decode": (C.c_int, [pp, vp, C.c_int]),
        "orc_point_encode_like_eddsa": (None, [vp, pp]),
        "orc_point_decode_like_eddsa": (C.c_int, [pp, vp]),
        "orc_point_scalarmul": (None, [pp, pp, ps]),
        "orc_precompute": (None, [C.POINTER(Precomputed), pp]),
        "orc_precompute_wnafs": (None, [C.POINTER(Niels), pp]),
        "orc_precomputed_scalarmul": (None, [pp, C.POINTER(Precomputed), ps]),
        "orc_point_double_scalarmul": (None, [pp, pp, ps, pp, ps]),
        "orc_base_double_scalarmul_non_secret": (None, [pp, ps, pp, ps]),
        "orc_direct_scalarmul": (C.c_int, [vp, vp, ps, C.c_int, C.c_int]),
        "orc_shake256": (None, [vp, C.c_size_t, vp, C.c_size_t]),
        "orc_point_from_hash_nonuniform": (None, [pp, vp]),
        "orc_point_from_hash_uniform": (None, [pp, vp]),
        "orc_x448": (C.c_int, [vp, vp, vp]),
        "orc_x448_derive_public_key": (None, [vp, vp]),
        "orc_point_encode_like_x448": (None, [vp, vp]),
        "orc_ed448_convert_public_key_to_x448": (None, [vp, vp]),
        "orc_ed448_derive_secret_scalar": (None, [vp, vp]),
        "orc_ed448_convert_private_key_to_x448": (None, [vp, vp]),
        "orc_ed448_derive_public_key": (None, [vp, vp]),
        "orc_ed448_sign": (None, [vp, vp, vp, vp, C.c_size_t, C.c_uint8, vp, C.c_uint8]),
        "orc_ed448_verify": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint8, vp, C.c_uint8]),
        "orc_point_scalarmul_batch": (None, [vp, vp, vp, C.c_size_t, C.c_int]),
        "orc_precomputed_scalarmul_batch": (None, [vp, vp, vp, C.c_size_t, C.c_int]),
        "orc_point_encode_batch": (None, [vp, vp, C.c_size_t, C.c_int]),
        "orc_extern_scalarmul_batch": (None, [vp, vp, vp, vp, C.c_size_t, C.c_int]),
        "orc_bench_extern_scalarmul": (None, [vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]),
        "orc_ed448_verify_batch": (None, [vp, vp, vp, vp, C.c_size_t, C.c_uint8, vp, C.c_uint8,
                                          C.c_size_t, C.c_int]),
        "orc_ed448_sign_batch": (None, [vp, vp, vp, vp, C.c_size_t, C.c_uint8, vp, C.c_uint8,
                                        C.c_size_t, C.c_int]),
        "orc_ed448_derive_public_key_batch": (None, [vp, vp, C.c_size_t, C.c_int]),
        "orc_point_double_scalarmul_batch": (None, [vp, vp, vp, vp, vp, C.c_size_t, C.c_int]),
        "orc_point_dual_scalarmul_batch": (None, [vp, vp, vp, vp, vp, C.c_size_t, C.c_int]),
        "orc_direct_scalarmul_batch": (None, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int]),
        "orc_base_table_entries": (None, [vp, C.c_uint, C.c_size_t, C.c_size_t, C.c_int]),
    }
    for name, (res, args) in proto.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _oracle = L
    return L


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The real reference (arch_ref64), only available in the build container."""
    global _ref
    if _ref is not None:
        return _ref
    L = C.CDLL(REF_SO)
    pp, ps = C.POINTER(Point), C.POINTER(Scalar)
    vp = C.c_void_p
    proto = {
        "goldilocks_448_point_scalarmul": (None, [pp, pp, ps]),
        "goldilocks_448_point_double_scalarmul": (None, [pp, pp, ps, pp, ps]),
        "goldilocks_448_base_double_scalarmul_non_secret": (None, [pp, ps, pp, ps]),
        "goldilocks_448_precomputed_scalarmul": (None, [pp, vp, ps]),
        "goldilocks_448_precompute": (None, [vp, pp]),
        "goldilocks_448_point_encode": (None, [vp, pp]),
        "goldilocks_448_point_decode": (C.c_int, [pp, vp, C.c_uint64]),
        "goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa": (None, [vp, pp]),
        "goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio": (C.c_int, [pp, vp]),
        "goldilocks_448_point_from_hash_uniform": (None, [pp, vp]),
        "goldilocks_448_point_from_hash_nonuniform": (None, [pp, vp]),
        "goldilocks_448_point_add": (None, [pp, pp, pp]),
        "goldilocks_448_point_sub": (None, [pp, pp, pp]),
        "goldilocks_448_point_double": (None, [pp, pp]),
        "goldilocks_448_point_debugging_torque": (None, [pp, pp]),
        "goldilocks_448_point_debugging_pscale": (None, [pp, pp, vp]),
        "goldilocks_448_point_eq": (C.c_uint64, [pp, pp]),
        "goldilocks_448_point_valid": (C.c_uint64, [pp]),
        "goldilocks_448_scalar_decode_long": (None, [ps, vp, C.c_size_t]),
        "goldilocks_448_scalar_decode": (C.c_int, [ps, vp]),
        "goldilocks_448_scalar_encode": (None, [vp, ps]),
        "goldilocks_448_scalar_add": (None, [ps, ps, ps]),
        "goldilocks_448_scalar_sub": (None, [ps, ps, ps]),
        "goldilocks_448_scalar_mul": (None, [ps, ps, ps]),
        "goldilocks_448_scalar_halve": (None, [ps, ps]),
        "goldilocks_448_scalar_invert": (C.c_int, [ps, ps]),
        "goldilocks_448_direct_scalarmul": (C.c_int, [vp, vp, ps, C.c_uint64, C.c_uint64]),
        "goldilocks_ed448_derive_public_key": (None, [vp, vp]),
        "goldilocks_ed448_sign": (None, [vp, vp, vp, vp, C.c_size_t, C.c_uint8, vp, C.c_uint8]),
        "goldilocks_ed448_verify": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_uint8, vp, C.c_uint8]),
        "goldilocks_sha3_hash": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, vp]),
        "goldilocks_x448": (C.c_int, [vp, vp, vp]),
        "goldilocks_x448_derive_public_key": (None, [vp, vp]),
        "goldilocks_448_point_mul_by_ratio_and_encode_like_x448": (None, [vp, vp]),
        "goldilocks_ed448_convert_public_key_to_x448": (None, [vp, vp]),
        "goldilocks_ed448_derive_secret_scalar": (None, [vp, vp]),
        "goldilocks_ed448_convert_private_key_to_x448": (None, [vp, vp]),
    }
    for name, (res, args) in proto.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L.point_base = Point.in_dll(L, "goldilocks_448_point_base")
    L.precomputed_base = C.c_void_p.in_dll(L, "goldilocks_448_precomputed_base")
    L.shake256_params = C.c_void_p(C.addressof(C.c_char.in_dll(L, "GOLDILOCKS_SHAKE256_params_s")))
    _ref = L
    return L


def buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(bytes(b))
