"""libgoldilocks_amd -- ctypes binding of the MI355X Ed448-Goldilocks batch engine.

This is the Python host-side mirror of the reference's own Python binding
(reference: python/edgold/ed448.py, a ctypes caller of libgoldilocks.so) extended with
the batch entry points of include/goldilocks_amd.h.  It contains no arithmetic: every
call goes through the C ABI of the in-tree ``libgoldilocks_amd.so`` (HIP, gfx950).  If
that library is missing the import fails loudly -- there is no fallback.

Array conventions (numpy, C-contiguous):
  points   uint64 [n, 32]   = goldilocks_448_point_s  (x,y,z,t, 8 x 56-bit limbs each)
  scalars  uint64 [n, 7]    = goldilocks_448_scalar_s (little-endian words, < q)
  tables   uint64 [1920]    = goldilocks_448_precomputed_s
  bytes    uint8  [n, 56|57|114]
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GOLDILOCKS_AMD_LIB selects an experimental build variant (tools/probes/gpu_probe.py); default: the in-tree product.
LIB_PATH = os.environ.get("GOLDILOCKS_AMD_LIB") or os.path.join(_HERE, "libgoldilocks_amd.so")

GOLDILOCKS_SUCCESS = -1
GOLDILOCKS_FAILURE = 0
SER_BYTES = 56
EDDSA_448_PUBLIC_BYTES = 57
EDDSA_448_SIGNATURE_BYTES = 114

# Every symbol include/goldilocks_amd.h declares (checked by tests/test_abi.py).
FUNCTIONS = {
    # (1) drop-in single ops
    "goldilocks_448_point_scalarmul": (None, "ppp"),
    "goldilocks_448_direct_scalarmul": (C.c_int, "pppQQ"),
    "goldilocks_448_precompute": (None, "pp"),
    "goldilocks_448_precomputed_scalarmul": (None, "ppp"),
    "goldilocks_448_point_double_scalarmul": (None, "ppppp"),
    "goldilocks_448_base_double_scalarmul_non_secret": (None, "pppp"),
    "goldilocks_448_point_encode": (None, "pp"),
    "goldilocks_448_point_decode": (C.c_int, "ppQ"),
    "goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa": (None, "pp"),
    "goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio": (C.c_int, "pp"),
    "goldilocks_448_point_eq": (C.c_uint64, "pp"),
    "goldilocks_448_point_valid": (C.c_uint64, "p"),
    "goldilocks_448_point_add": (None, "ppp"),
    "goldilocks_448_point_sub": (None, "ppp"),
    "goldilocks_448_point_double": (None, "pp"),
    "goldilocks_448_point_negate": (None, "pp"),
    "goldilocks_448_point_debugging_torque": (None, "pp"),
    "goldilocks_448_point_debugging_pscale": (None, "ppp"),
    "goldilocks_448_scalar_decode": (C.c_int, "pp"),
    "goldilocks_448_scalar_decode_long": (None, "ppz"),
    "goldilocks_448_scalar_encode": (None, "pp"),
    "goldilocks_448_scalar_add": (None, "ppp"),
    "goldilocks_448_scalar_sub": (None, "ppp"),
    "goldilocks_448_scalar_mul": (None, "ppp"),
    "goldilocks_448_scalar_halve": (None, "pp"),
    "goldilocks_448_scalar_invert": (C.c_int, "pp"),
    "goldilocks_448_scalar_eq": (C.c_uint64, "pp"),
    "goldilocks_448_scalar_set_unsigned": (None, "pQ"),
    "goldilocks_448_scalar_cond_sel": (None, "pppQ"),
    "goldilocks_448_scalar_destroy": (None, "p"),
    "goldilocks_448_point_cond_sel": (None, "pppQ"),
    "goldilocks_448_point_destroy": (None, "p"),
    "goldilocks_448_precomputed_destroy": (None, "p"),
    "goldilocks_ed448_verify": (C.c_int, "pppzBpB"),
    "goldilocks_ed448_derive_public_key": (None, "pp"),
    "goldilocks_448_point_dual_scalarmul": (None, "ppppp"),
    "goldilocks_448_point_from_hash_nonuniform": (None, "pp"),
    "goldilocks_448_point_from_hash_uniform": (None, "pp"),
    "goldilocks_x448": (C.c_int, "ppp"),
    "goldilocks_x448_derive_public_key": (None, "pp"),
    "goldilocks_ed448_sign": (None, "ppppzBpB"),
    # (2) host-array batches
    "goldilocks_448_point_scalarmul_batch": (C.c_int, "pppz"),
    "goldilocks_448_precomputed_scalarmul_batch": (C.c_int, "pppz"),
    "goldilocks_448_point_scalarmul_batch_ex": (C.c_int, "pppzIpi"),
    "goldilocks_448_precomputed_scalarmul_batch_ex": (C.c_int, "pppzIpi"),
    "goldilocks_ed448_verify_batch_ex": (C.c_int, "pppppBpBzpi"),
    "goldilocks_448_point_double_scalarmul_batch": (C.c_int, "pppppz"),
    "goldilocks_448_point_encode_batch": (C.c_int, "ppz"),
    "goldilocks_448_point_decode_batch": (C.c_int, "pppQz"),
    "goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa_batch": (C.c_int, "ppz"),
    "goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio_batch": (C.c_int, "pppz"),
    "goldilocks_ed448_verify_batch": (C.c_int, "pppppBpBz"),
    "goldilocks_ed448_derive_public_key_batch": (C.c_int, "ppz"),
    "goldilocks_ed448_sign_batch": (C.c_int, "pppppBpBz"),
    "goldilocks_448_direct_scalarmul_batch": (C.c_int, "ppppQQz"),
    "goldilocks_x448_batch": (C.c_int, "ppppz"),
    "goldilocks_448_point_mul_by_ratio_and_encode_like_x448_batch": (C.c_int, "ppz"),
    "goldilocks_ed448_convert_public_key_to_x448_batch": (C.c_int, "ppz"),
    "goldilocks_ed448_derive_secret_scalar_batch": (C.c_int, "ppz"),
    "goldilocks_ed448_convert_private_key_to_x448_batch": (C.c_int, "ppz"),
    "goldilocks_448_point_mul_by_ratio_and_encode_like_x448": (None, "pp"),
    "goldilocks_ed448_convert_public_key_to_x448": (None, "pp"),
    "goldilocks_ed448_derive_secret_scalar": (None, "pp"),
    "goldilocks_ed448_convert_private_key_to_x448": (None, "pp"),
    "goldilocks_448_point_dual_scalarmul_batch": (C.c_int, "pppppz"),
    "goldilocks_448_point_from_hash_batch": (C.c_int, "ppiz"),
    # (3) device-array API
    "goldilocks_amd_init": (C.c_int, "i"),
    "goldilocks_amd_shutdown": (None, ""),
    "goldilocks_amd_last_error": (C.c_char_p, ""),
    "goldilocks_amd_build_info": (C.c_char_p, ""),
    "goldilocks_amd_device_info": (C.c_int, "pzpp"),
    "goldilocks_amd_use_devices": (C.c_int, "pi"),
    "goldilocks_amd_set_table_access": (C.c_int, "i"),
    "goldilocks_amd_get_table_access": (C.c_int, ""),
    "goldilocks_amd_thread_mode_counts": (None, "p"),
    "goldilocks_amd_last_verify_key_counts": (C.c_int, "p"),
    "goldilocks_amd_base_table_export": (C.c_int, "pQz"),
    "goldilocks_amd_set_wave_batch_max": (None, "z"),
    "goldilocks_amd_get_wave_batch_max": (C.c_size_t, ""),
    "goldilocks_amd_set_verify_key_pool": (None, "zz"),
    "goldilocks_amd_set_verify_key_combs": (None, "zz"),
    "goldilocks_amd_set_verify_key_combs_wide": (None, "z"),
    "goldilocks_amd_set_verify_key_combs_bytes": (None, "z"),
    "goldilocks_amd_set_verify_key_combs_xwide": (None, "z"),
    "goldilocks_amd_set_base_table_bits": (C.c_int, "i"),
    "goldilocks_amd_get_base_table_bits": (C.c_int, ""),
    "goldilocks_amd_release_memory": (C.c_int, "I"),
    "goldilocks_amd_point_scalarmul_dev": (C.c_int, "pppzp"),
    "goldilocks_amd_precomputed_scalarmul_dev": (C.c_int, "pppzp"),
    "goldilocks_amd_point_double_scalarmul_dev": (C.c_int, "pppppzp"),
    "goldilocks_amd_base_double_scalarmul_dev": (C.c_int, "ppppzp"),
    "goldilocks_amd_point_encode_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_point_decode_dev": (C.c_int, "pppizp"),
    "goldilocks_amd_point_encode_eddsa_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_point_decode_eddsa_dev": (C.c_int, "pppzp"),
    "goldilocks_amd_point_op_dev": (C.c_int, "pppizp"),
    "goldilocks_amd_scalar_op_dev": (C.c_int, "ppppizzp"),
    "goldilocks_amd_scalar_op_batch": (C.c_int, "ppppizz"),
    "goldilocks_amd_point_pred_dev": (C.c_int, "pppizp"),
    "goldilocks_amd_precompute_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_ed448_verify_dev": (C.c_int, "pppppzBpBzp"),
    "goldilocks_amd_field_op_dev": (C.c_int, "ppppizp"),
    "goldilocks_amd_wave_field_op_dev": (C.c_int, "ppppizp"),
    "goldilocks_amd_half_size_pair_dev": (C.c_int, "pppzp"),
    "goldilocks_amd_ed448_derive_public_key_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_ed448_sign_dev": (C.c_int, "pppppzBpBzp"),
    "goldilocks_amd_direct_scalarmul_dev": (C.c_int, "ppppiizp"),
    "goldilocks_amd_x448_dev": (C.c_int, "ppppzp"),
    "goldilocks_amd_point_encode_like_x448_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_ed448_convert_public_key_to_x448_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_ed448_derive_secret_scalar_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_ed448_convert_private_key_to_x448_dev": (C.c_int, "ppzp"),
    "goldilocks_amd_point_dual_scalarmul_dev": (C.c_int, "pppppzp"),
    "goldilocks_amd_point_from_hash_dev": (C.c_int, "ppizp"),
    # ... with the table access of the call (GOLDILOCKS_AMD_CALL_TABLES_*)
    "goldilocks_amd_point_scalarmul_dev_ex": (C.c_int, "pppzpI"),
    "goldilocks_amd_precomputed_scalarmul_dev_ex": (C.c_int, "pppzpI"),
    "goldilocks_amd_point_double_scalarmul_dev_ex": (C.c_int, "pppppzpI"),
    "goldilocks_amd_ed448_derive_public_key_dev_ex": (C.c_int, "ppzpI"),
    "goldilocks_amd_ed448_sign_dev_ex": (C.c_int, "pppppzBpBzpI"),
    "goldilocks_amd_direct_scalarmul_dev_ex": (C.c_int, "ppppiizpI"),
    "goldilocks_amd_point_dual_scalarmul_dev_ex": (C.c_int, "pppppzpI"),
    "goldilocks_amd_x448_dev_ex": (C.c_int, "ppppzpI"),
}
DATA_SYMBOLS = [
    "goldilocks_448_sizeof_precomputed_s", "goldilocks_448_alignof_precomputed_s",
    "goldilocks_448_scalar_one", "goldilocks_448_scalar_zero", "goldilocks_448_point_identity",
    "goldilocks_448_point_base", "goldilocks_448_precomputed_base", "goldilocks_x448_base_point",
]
_CT = {"p": C.c_void_p, "z": C.c_size_t, "Q": C.c_uint64, "B": C.c_uint8, "i": C.c_int, "I": C.c_uint32}

_lib = None


class GoldilocksAmdError(RuntimeError):
    pass


def lib():
    """The loaded libgoldilocks_amd.so (raises ImportError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64; when torch is
        # going to be used for device memory it must be the first to load it, otherwise the two
        # runtimes fight over the device ("No HIP GPUs are available").  Plumbing only.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, sig) in FUNCTIONS.items():
            try:
                f = getattr(L, name)
            except AttributeError:
                # an experimental build selected with GOLDILOCKS_AMD_LIB may predate an entry point (A/B runs against an
                # older round's library); the product must export everything the header declares (tests/test_abi.py)
                if "GOLDILOCKS_AMD_LIB" in os.environ:
                    continue
                raise
            f.restype = res
            f.argtypes = [_CT[ch] for ch in sig]
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise GoldilocksAmdError(lib().goldilocks_amd_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    assert a.shape[1] == cols, a.shape
    return a


def _u8(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    assert a.shape[1] == cols, a.shape
    return a


# ----------------------------------------------------------------------------- constants


def point_base():
    return np.ctypeslib.as_array((C.c_uint64 * 32).in_dll(lib(), "goldilocks_448_point_base")).copy()


def point_identity():
    return np.ctypeslib.as_array((C.c_uint64 * 32).in_dll(lib(), "goldilocks_448_point_identity")).copy()


def precomputed_base():
    """The built-in base-point comb table (copy of the 15360 bytes the exported pointer refers to)."""
    p = C.c_void_p.in_dll(lib(), "goldilocks_448_precomputed_base")
    return np.frombuffer(C.string_at(p.value, 15360), dtype=np.uint64).copy()


# ----------------------------------------------------------------------------- host-array batches


CALL_TABLES_DEFAULT, CALL_TABLES_FAST, CALL_TABLES_INDEX_INDEPENDENT = 0, 1, 2   # the `flags` of the *_ex entry points


def _devs(devices):
    devices = list(devices or [])
    arr = (C.c_int * max(1, len(devices)))(*devices)
    return arr, (C.addressof(arr) if devices else None), len(devices)


def point_scalarmul_batch(bases, scalars, flags=CALL_TABLES_DEFAULT, devices=None):
    """flags / devices: the table access and the GPUs of THIS call (goldilocks_448_point_scalarmul_batch_ex)."""
    bases, scalars = _u64(bases, 32), _u64(scalars, 7)
    n = len(scalars)
    assert len(bases) == n
    out = np.empty((n, 32), dtype=np.uint64)
    keep, dp, dn = _devs(devices)
    _check(lib().goldilocks_448_point_scalarmul_batch_ex(_ptr(out), _ptr(bases), _ptr(scalars), n, flags, dp, dn))
    return out


def precomputed_scalarmul_batch(scalars, table=None, flags=CALL_TABLES_DEFAULT, devices=None):
    scalars = _u64(scalars, 7)
    n = len(scalars)
    out = np.empty((n, 32), dtype=np.uint64)
    if table is None:
        tab = C.c_void_p.in_dll(lib(), "goldilocks_448_precomputed_base")
    else:
        table = np.ascontiguousarray(table, dtype=np.uint64).reshape(1920)
        tab = _ptr(table)
    keep, dp, dn = _devs(devices)
    _check(lib().goldilocks_448_precomputed_scalarmul_batch_ex(_ptr(out), tab, _ptr(scalars), n, flags, dp, dn))
    return out


def point_double_scalarmul_batch(bases1, scalars1, bases2, scalars2):
    """scalars1*bases1 + scalars2*bases2; bases1=None means the base point for every lane."""
    scalars1, bases2, scalars2 = _u64(scalars1, 7), _u64(bases2, 32), _u64(scalars2, 7)
    n = len(scalars1)
    out = np.empty((n, 32), dtype=np.uint64)
    b1 = None if bases1 is None else _ptr(_u64(bases1, 32))
    _check(lib().goldilocks_448_point_double_scalarmul_batch(_ptr(out), b1, _ptr(scalars1), _ptr(bases2),
                                                             _ptr(scalars2), n))
    return out


def point_encode_batch(points):
    points = _u64(points, 32)
    out = np.empty((len(points), 56), dtype=np.uint8)
    _check(lib().goldilocks_448_point_encode_batch(_ptr(out), _ptr(points), len(points)))
    return out


def point_decode_batch(ser, allow_identity=False):
    ser = _u8(ser, 56)
    n = len(ser)
    pts = np.empty((n, 32), dtype=np.uint64)
    st = np.empty(n, dtype=np.int32)
    _check(lib().goldilocks_448_point_decode_batch(_ptr(pts), _ptr(st), _ptr(ser),
                                                   2**64 - 1 if allow_identity else 0, n))
    return pts, st


def point_encode_like_eddsa_batch(points):
    points = _u64(points, 32)
    out = np.empty((len(points), 57), dtype=np.uint8)
    _check(lib().goldilocks_448_point_mul_by_ratio_and_encode_like_eddsa_batch(_ptr(out), _ptr(points),
                                                                               len(points)))
    return out


def point_decode_like_eddsa_batch(enc):
    enc = _u8(enc, 57)
    n = len(enc)
    pts = np.empty((n, 32), dtype=np.uint64)
    st = np.empty(n, dtype=np.int32)
    _check(lib().goldilocks_448_point_decode_like_eddsa_and_mul_by_ratio_batch(_ptr(pts), _ptr(st), _ptr(enc), n))
    return pts, st


def ed448_verify_batch(sigs, pks, messages, prehashed=False, context=b"", devices=None):
    """status[i] in {-1 (valid), 0 (invalid)} for each (sig, pk, message); devices: the GPUs of this call."""
    sigs, pks = _u8(sigs, 114), _u8(pks, 57)
    n = len(sigs)
    assert len(pks) == n and len(messages) == n
    bufs = [C.create_string_buffer(bytes(m), max(len(m), 1)) for m in messages]
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * n)(*[len(m) for m in messages])
    ctx = C.create_string_buffer(bytes(context), max(len(context), 1))
    st = np.empty(n, dtype=np.int32)
    keep, dp, dn = _devs(devices)
    _check(lib().goldilocks_ed448_verify_batch_ex(_ptr(st), _ptr(sigs), _ptr(pks), C.addressof(ptrs),
                                                  C.addressof(lens), 1 if prehashed else 0, C.addressof(ctx),
                                                  len(context), n, dp, dn))
    return st


def _msg_tables(messages):
    n = len(messages)
    bufs = [C.create_string_buffer(bytes(m), max(len(m), 1)) for m in messages]
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * n)(*[len(m) for m in messages])
    return bufs, ptrs, lens


def ed448_derive_public_key_batch(sks):
    sks = _u8(sks, 57)
    out = np.empty((len(sks), 57), dtype=np.uint8)
    _check(lib().goldilocks_ed448_derive_public_key_batch(_ptr(out), _ptr(sks), len(sks)))
    return out


def ed448_sign_batch(sks, pks, messages, prehashed=False, context=b""):
    sks, pks = _u8(sks, 57), _u8(pks, 57)
    n = len(sks)
    assert len(pks) == n and len(messages) == n
    bufs, ptrs, lens = _msg_tables(messages)
    ctx = C.create_string_buffer(bytes(context), max(len(context), 1))
    out = np.empty((n, 114), dtype=np.uint8)
    _check(lib().goldilocks_ed448_sign_batch(_ptr(out), _ptr(sks), _ptr(pks), C.addressof(ptrs), C.addressof(lens),
                                             1 if prehashed else 0, C.addressof(ctx), len(context), n))
    return out


def direct_scalarmul_batch(bases56, scalars, allow_identity=False, short_circuit=False):
    """Wire-format scalarmul: 56-byte encodings in, 56-byte encodings + status out."""
    bases56, scalars = _u8(bases56, 56), _u64(scalars, 7)
    n = len(scalars)
    out = np.zeros((n, 56), dtype=np.uint8)
    st = np.empty(n, dtype=np.int32)
    _check(lib().goldilocks_448_direct_scalarmul_batch(_ptr(out), _ptr(st), _ptr(bases56), _ptr(scalars),
                                                       2**64 - 1 if allow_identity else 0,
                                                       2**64 - 1 if short_circuit else 0, n))
    return out, st


def point_dual_scalarmul_batch(bases, scalars1, scalars2):
    bases, scalars1, scalars2 = _u64(bases, 32), _u64(scalars1, 7), _u64(scalars2, 7)
    n = len(bases)
    o1, o2 = np.empty((n, 32), dtype=np.uint64), np.empty((n, 32), dtype=np.uint64)
    _check(lib().goldilocks_448_point_dual_scalarmul_batch(_ptr(o1), _ptr(o2), _ptr(bases), _ptr(scalars1),
                                                           _ptr(scalars2), n))
    return o1, o2


def point_from_hash_batch(hashes, uniform=False):
    """Elligator 2: uint8 [n, 56] (nonuniform) or [n, 112] (uniform) -> points."""
    hashes = _u8(hashes, 112 if uniform else 56)
    out = np.empty((len(hashes), 32), dtype=np.uint64)
    _check(lib().goldilocks_448_point_from_hash_batch(_ptr(out), _ptr(hashes), 1 if uniform else 0, len(hashes)))
    return out


def x448_batch(scalars56, bases56=None):
    """X448(scalar, base) for each row; bases56=None computes public keys (base point 5)."""
    scalars56 = _u8(scalars56, 56)
    n = len(scalars56)
    out = np.empty((n, 56), dtype=np.uint8)
    st = np.empty(n, dtype=np.int32)
    b = None if bases56 is None else _ptr(_u8(bases56, 56))
    _check(lib().goldilocks_x448_batch(_ptr(out), _ptr(st), b, _ptr(scalars56), n))
    return out, st


def x448_from_edwards_batch(kind, rows):
    """The rest of the reference's X448 surface over host arrays, 56 bytes out per row.  kind: "point" (256-byte points ->
    point_mul_by_ratio_and_encode_like_x448), "public" (57-byte Ed448 public keys -> convert_public_key_to_x448),
    "private" (57-byte Ed448 private keys -> convert_private_key_to_x448), "scalar" (private keys -> derive_secret_scalar)."""
    width, fn = {"point": (256, lib().goldilocks_448_point_mul_by_ratio_and_encode_like_x448_batch),
                 "public": (57, lib().goldilocks_ed448_convert_public_key_to_x448_batch),
                 "private": (57, lib().goldilocks_ed448_convert_private_key_to_x448_batch),
                 "scalar": (57, lib().goldilocks_ed448_derive_secret_scalar_batch)}[kind]
    rows = _u8(rows, width)
    out = np.empty((len(rows), 56), dtype=np.uint8)
    _check(fn(_ptr(out), _ptr(rows), len(rows)))
    return out


SCALAR_OPS = {"add": 0, "sub": 1, "mul": 2, "halve": 3, "invert": 4, "decode": 5, "decode_long": 6}


def scalar_op_batch(op, a, b=None, length=0):
    """The reference's scalar API over host arrays (src/scalar.c; goldilocks_amd_scalar_op_batch).  op: a key of SCALAR_OPS.
    a, b: [n, 7] uint64 scalars (add, sub, mul, halve, invert); decode: a = [n, 56] bytes; decode_long: a = n * length bytes.
    Returns the [n, 7] results; for invert and decode a pair (results, int32 status: -1 success / 0 failure)."""
    code = SCALAR_OPS[op]
    if code <= 4:
        a = _u64(a, 7)
        n = len(a)
    elif code == 5:
        a = _u8(a, 56)
        n = len(a)
    else:
        if length <= 0:
            raise ValueError("decode_long: the strings' common length in bytes")
        a = np.ascontiguousarray(np.frombuffer(bytes(a), dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8).reshape(-1)
        n = len(a) // length
    if code <= 2:
        b = _u64(b, 7)
        if len(b) != n:
            raise ValueError("operand arrays differ in length")
    out = np.empty((n, 7), dtype=np.uint64)
    st = np.empty(n, dtype=np.int32)
    _check(lib().goldilocks_amd_scalar_op_batch(_ptr(out), _ptr(st) if code in (4, 5) else None, _ptr(a),
                                                 _ptr(b) if code <= 2 else None, code, length, n))
    return (out, st) if code in (4, 5) else out


# ----------------------------------------------------------------------------- single ops (drop-in names)


def point_scalarmul(base, scalar):
    base, scalar = _u64(base, 32), _u64(scalar, 7)
    out = np.empty((1, 32), dtype=np.uint64)
    lib().goldilocks_448_point_scalarmul(_ptr(out), _ptr(base), _ptr(scalar))
    return out[0]


def precompute(base):
    base = _u64(base, 32)
    tab = np.empty(1920, dtype=np.uint64)
    lib().goldilocks_448_precompute(_ptr(tab), _ptr(base))
    return tab


def ed448_verify(sig, pk, msg, context=b"", prehashed=False):
    sig, pk = _u8(np.frombuffer(bytes(sig), np.uint8), 114), _u8(np.frombuffer(bytes(pk), np.uint8), 57)
    m = C.create_string_buffer(bytes(msg), max(len(msg), 1))
    ctx = C.create_string_buffer(bytes(context), max(len(context), 1))
    return lib().goldilocks_ed448_verify(_ptr(sig), _ptr(pk), C.addressof(m), len(msg), 1 if prehashed else 0,
                                         C.addressof(ctx), len(context)) == GOLDILOCKS_SUCCESS


class EDDSA448(object):
    """Verify-only mirror of the reference binding's class (python/edgold/ed448.py:109-201)."""

    def __init__(self, pub):
        if len(pub) != EDDSA_448_PUBLIC_BYTES:
            raise ValueError("public key must be 57 bytes")
        self._pub = bytes(pub)

    def verify(self, sig, msg, ctx=None):
        """Raises ValueError if sig is not valid for msg (same behaviour as the reference)."""
        if not ed448_verify(sig, self._pub, msg, context=ctx or b""):
            raise ValueError("signature is not valid")


# ----------------------------------------------------------------------------- device-array API (torch / raw pointers)


def dev(name, *args, flags=None):
    """Call goldilocks_amd_<name>_dev with raw device pointers (ints) / sizes; raises on error.
    flags (CALL_TABLES_*): the table access of this call, through goldilocks_amd_<name>_dev_ex."""
    if flags is None:
        _check(getattr(lib(), "goldilocks_amd_%s_dev" % name)(*args))
    else:
        _check(getattr(lib(), "goldilocks_amd_%s_dev_ex" % name)(*args, flags))


def use_devices(devices=None):
    """Shard the host-array batches (variable-base, fixed-base, verify) over these HIP devices, one
    host thread per device; None or [] restores the default (current device only)."""
    devices = list(devices or [])
    arr = (C.c_int * max(1, len(devices)))(*devices)
    _check(lib().goldilocks_amd_use_devices(C.addressof(arr) if devices else None, len(devices)))


TABLES_FAST, TABLES_INDEX_INDEPENDENT = 0, 1


def set_table_access(mode):
    """TABLES_INDEX_INDEPENDENT (the library's default, the reference's constant-time contract) or
    TABLES_FAST (opt-in for public scalars): how every kernel whose scalar may be secret looks up its
    window / comb table -- see include/goldilocks_amd.h."""
    _check(lib().goldilocks_amd_set_table_access(int(mode)))


def set_wave_batch_max(n):
    """Batches of up to n variable-base multiplications run one operation per wavefront (0 disables)."""
    lib().goldilocks_amd_set_wave_batch_max(int(n))


def get_wave_batch_max():
    return lib().goldilocks_amd_get_wave_batch_max()


KEY_POOL_DEFAULT, KEY_POOL_MIN_BATCH_DEFAULT = 1 << 18, 1 << 16


def set_verify_key_pool(keys=KEY_POOL_DEFAULT, min_batch=KEY_POOL_MIN_BATCH_DEFAULT):
    """Verification shares one decoding and one window table between the signatures of a key: a pool of `keys`
    tables for batches of at least `min_batch` signatures (0 keys turns it off)."""
    lib().goldilocks_amd_set_verify_key_pool(int(keys), int(min_batch))


KEY_COMBS_DEFAULT, KEY_COMBS_MIN_PER_KEY_DEFAULT = 1 << 17, 8


def set_verify_key_combs(keys=KEY_COMBS_DEFAULT, min_signatures_per_key=KEY_COMBS_MIN_PER_KEY_DEFAULT):
    """Keys that sign at least `min_signatures_per_key` signatures of a batch on average (twice that below 2^18
    signatures) get a fixed-base comb each (at most `keys` of them; 0 turns the combs off)."""
    lib().goldilocks_amd_set_verify_key_combs(int(keys), int(min_signatures_per_key))


KEY_COMBS_WIDE_MIN_PER_KEY_DEFAULT = 256


def set_verify_key_combs_wide(min_signatures_per_key=KEY_COMBS_WIDE_MIN_PER_KEY_DEFAULT):
    """Keys that sign at least so many signatures of a batch on average get the wider comb (8 teeth); 0: never."""
    lib().goldilocks_amd_set_verify_key_combs_wide(int(min_signatures_per_key))


KEY_COMBS_XWIDE_MIN_PER_KEY_DEFAULT = 1024


def set_verify_key_combs_xwide(min_signatures_per_key=KEY_COMBS_XWIDE_MIN_PER_KEY_DEFAULT):
    """Keys that sign at least so many signatures of a batch on average get the widest comb (5 x 9 teeth); 0: never."""
    lib().goldilocks_amd_set_verify_key_combs_xwide(int(min_signatures_per_key))


BASE_TABLE_BITS_DEFAULT, BASE_TABLE_BITS_AUTO = 20, 1


def set_verify_key_combs_bytes(nbytes=4 << 30):
    """Ceiling on the workspace a verification reserves for per-key combs (71 KiB per key)."""
    lib().goldilocks_amd_set_verify_key_combs_bytes(int(nbytes))


def set_base_table_bits(bits=0):
    """Digit width of the base point's window table: 0 = the default (20 bits, 2.2 GiB), BASE_TABLE_BITS_AUTO = the widest
    that takes at most an eighth of the device's free memory, or an even width from 8 to 24."""
    if lib().goldilocks_amd_set_base_table_bits(int(bits)) != 0:
        raise ValueError(lib().goldilocks_amd_last_error().decode())


def get_base_table_bits():
    """Digit width of the table the current device holds (0: none built yet)."""
    return lib().goldilocks_amd_get_base_table_bits()


RELEASE_WORKSPACE, RELEASE_STAGING, RELEASE_BASE_TABLE, RELEASE_ALL = 1, 2, 4, 7


def release_memory(what=RELEASE_ALL):
    """Give the current device's workspace / staging buffers / base-point window table back (they come back on demand)."""
    _check(lib().goldilocks_amd_release_memory(int(what)))


def get_table_access():
    return lib().goldilocks_amd_get_table_access()


def thread_mode_counts():
    """(fast, index-independent): how the mode-dependent calls of THIS thread resolved so far (test hook)."""
    c = (C.c_uint64 * 2)()
    lib().goldilocks_amd_thread_mode_counts(C.addressof(c))
    return int(c[0]), int(c[1])


def last_verify_key_counts(teeth=False):
    """(distinct keys, keys with a pooled window table, keys with a comb) of the last large verification batch on this
    device (test hook); teeth=True: the combs' teeth (7, 8, or 0 without combs) as a fourth number."""
    c = (C.c_uint32 * 4)()
    _check(lib().goldilocks_amd_last_verify_key_counts(C.addressof(c)))
    return (int(c[0]), int(c[1]), int(c[2]), int(c[3])) if teeth else (int(c[0]), int(c[1]), int(c[2]))


def build_info():
    """{"toolchain": what compiled the loaded library, "library_sha256": the digest of the file that was loaded}: the stamp
    that bench lines, GPU test logs and soak logs carry (parity evidence belongs to one code object of one toolchain)."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    try:
        toolchain = lib().goldilocks_amd_build_info().decode()
    except AttributeError:      # an older library loaded as an A/B variant (GOLDILOCKS_AMD_LIB)
        toolchain = None
    return {"toolchain": toolchain, "library_sha256": digest}


def device_info():
    arch = C.create_string_buffer(64)
    cus = C.c_int()
    ws = C.c_size_t()
    _check(lib().goldilocks_amd_device_info(C.addressof(arch), 64, C.addressof(cus), C.addressof(ws)))
    return {"arch": arch.value.decode(), "compute_units": cus.value, "workspace_bytes": ws.value}
