"""GPU tests of the base point's window table at every digit width the library offers
(goldilocks_amd_set_base_table_bits; scalarmul.hpp ladder_bwt, kernels_fixed.hip k_build_bwt).  The width is a
memory / additions trade and must not show in any result: what multiplies the base point through the table --
S*B of a verification (reference: src/eddsa.c:283-300), base_double_scalarmul_non_secret (src/goldilocks.c:1145-1210),
and with digit-addressed tables key derivation, signing and the base point's precomputed_scalarmul -- is checked
against the oracle and the reference's golden fixtures at widths whose tables differ in everything else: span of the
recoding (448 / 450 / 460 bits), digits that straddle words, a top digit that reaches past bit 448."""
import json
import os

import numpy as np
import pytest

import _gen

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WIDTHS = [8, 16, 18, 20]          # 1.3 MiB, 168 MiB, 600 MiB, 2.2 GiB (22 and 24 bits: the default on an empty device, below)


@pytest.fixture()
def width(ga):
    """Sets a width for the body and leaves the library on its default (20 bits) afterwards."""
    def set_(bits):
        ga.set_base_table_bits(bits)
    try:
        yield set_
    finally:
        ga.set_base_table_bits(0)


def _f3_groups():
    cases = json.load(open(os.path.join(GOLD, "f3_verify.json")))["cases"]
    groups = {}
    for c in cases:
        groups.setdefault((c["ctx"], c["prehashed"]), []).append(c)
    return groups


@pytest.mark.parametrize("bits", WIDTHS)
def test_every_user_of_the_table_at_this_width(ga, O, width, bits):
    from _libs import Q
    width(bits)
    n = 300
    s = _gen.stream_scalars(n, b"bt/s/%d" % bits)
    s[:8] = _gen.scalars_from_ints([0, 1, Q - 1, 2**445, 2**444 - 1, 2, 255, 2**224])
    want = _gen.oracle_encode(_gen.oracle_fixed(O, s))
    # the base point's multiplication with digit-addressed tables, lane kernel (n > the wave path's reach is not needed:
    # both take the table) and through the per-call flag
    got = ga.point_encode_batch(ga.precomputed_scalarmul_batch(s, flags=ga.CALL_TABLES_FAST))
    assert ga.get_base_table_bits() == bits
    assert (got == want).all()
    big = np.concatenate([s] * 40)                 # 12 000: the lane kernel
    got = ga.point_encode_batch(ga.precomputed_scalarmul_batch(big, flags=ga.CALL_TABLES_FAST))
    assert (got == np.concatenate([want] * 40)).all()
    # s1 * B + s2 * P with the base point's half through the table (wave and lane kernels)
    for m in (5, n):
        k = _gen.stream_scalars(m, b"bt/p/%d" % bits)
        pts = _gen.oracle_fixed(O, k)
        s2 = _gen.stream_scalars(m, b"bt/s2/%d" % bits)
        got = ga.point_encode_batch(ga.point_double_scalarmul_batch(None, s[:m], pts, s2))
        ref = (np.array([int.from_bytes(x.tobytes(), "little") for x in s[:m]], dtype=object) +
               np.array([int.from_bytes(x.tobytes(), "little") for x in k], dtype=object) *
               np.array([int.from_bytes(x.tobytes(), "little") for x in s2], dtype=object)) % Q
        assert (got == _gen.oracle_encode(_gen.oracle_fixed(O, _gen.scalars_from_ints(list(ref))))).all(), m
    # key derivation and signatures with digit-addressed tables: RFC 8032's bytes whatever the width
    sks = np.frombuffer(_gen.stream(b"bt/sk/%d" % bits, 57 * 70), np.uint8).reshape(70, 57)
    ga.set_table_access(ga.TABLES_FAST)
    try:
        pks = ga.ed448_derive_public_key_batch(sks)
        msgs = [_gen.stream(b"bt/m%d" % i, i) for i in range(70)]
        sigs = ga.ed448_sign_batch(sks, pks, msgs)
    finally:
        ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT)
    assert (pks == ga.ed448_derive_public_key_batch(sks)).all()          # the default path uses no window table
    assert (sigs == ga.ed448_sign_batch(sks, pks, msgs)).all()
    assert list(ga.ed448_verify_batch(sigs, pks, msgs)) == [-1] * 70
    # verification: the reference's 256 golden cases (wave kernel), and replicated into batches the lane kernels take
    for (ctx, ph), cs in _f3_groups().items():
        sigs = np.array([np.frombuffer(bytes.fromhex(c["sig"]), np.uint8) for c in cs])
        pks = np.array([np.frombuffer(bytes.fromhex(c["pk"]), np.uint8) for c in cs])
        msgs = [bytes.fromhex(c["msg"]) for c in cs]
        verdicts = [c["verdict"] for c in cs]
        assert list(ga.ed448_verify_batch(sigs, pks, msgs, prehashed=bool(ph), context=bytes.fromhex(ctx))) == verdicts
        reps = -(-8192 // len(cs))
        order = np.random.default_rng(bits).permutation(len(cs) * reps) % len(cs)
        got = np.asarray(ga.ed448_verify_batch(sigs[order], pks[order], [msgs[i] for i in order], prehashed=bool(ph),
                                               context=bytes.fromhex(ctx)))
        assert (got == np.array(verdicts)[order]).all()


def test_default_width_is_fixed_auto_follows_the_free_memory_and_bad_widths_are_refused(ga, O, width):
    """The default is 20 bits (2.2 GiB) whatever the device holds; GOLDILOCKS_AMD_BASE_TABLE_BITS_AUTO asks for the widest
    table within an eighth of the free memory (the default until round 4)."""
    import torch
    s = _gen.stream_scalars(64, b"bt/default")
    want = _gen.oracle_encode(_gen.oracle_fixed(O, s))
    width(0)
    assert (ga.point_encode_batch(ga.precomputed_scalarmul_batch(s, flags=ga.CALL_TABLES_FAST)) == want).all()
    assert ga.get_base_table_bits() == ga.BASE_TABLE_BITS_DEFAULT == 20
    ga.release_memory(ga.RELEASE_BASE_TABLE)
    width(ga.BASE_TABLE_BITS_AUTO)
    assert (ga.point_encode_batch(ga.precomputed_scalarmul_batch(s, flags=ga.CALL_TABLES_FAST)) == want).all()
    free, _total = torch.cuda.mem_get_info()
    bits = ga.get_base_table_bits()
    assert bits in (16, 18, 20, 22, 24)
    entries = -(-446 // bits) << (bits - 1)
    assert bits == 16 or entries * 192 <= (free + entries * 192) // 8 + (1 << 30)     # an eighth of what was free before it
    for bad in (7, 9, 26, -2, 17):
        with pytest.raises(ValueError):
            ga.set_base_table_bits(bad)
    assert ga.get_base_table_bits() == bits          # (a refused width changes nothing)
    ga.release_memory(ga.RELEASE_BASE_TABLE)


def test_default_footprint_after_a_full_size_verification_is_bounded(ga, O, width):
    """include/goldilocks_amd.h documents what the library holds on a device at its default settings: 2.2 GiB of base-point
    table and a workspace of at most 7.5 GiB after 2^20 verifications (config 4's batch: 2^10 keys x 1 024 signatures) --
    10 GiB in all, whatever else is free on the device."""
    import torch
    width(0)
    ga.release_memory()
    n = 1 << 20
    sigs, pks, msgs = _gen.signatures(O, 1 << 12, msglen=32, seed=b"bt/footprint", nkeys=1 << 10)
    reps = n // len(sigs)
    dsig = torch.from_numpy(np.concatenate([sigs] * reps)).cuda()
    dpk = torch.from_numpy(np.concatenate([pks] * reps)).cuda()
    dmsg = torch.from_numpy(np.concatenate([np.frombuffer(b"".join(msgs), dtype=np.uint8)] * reps)).cuda()
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("ed448_verify", st.data_ptr(), dsig.data_ptr(), dpk.data_ptr(), dmsg.data_ptr(), None, 32, 0, None, 0, n, None)
    assert int((st == -1).sum()) == n
    assert ga.get_base_table_bits() == 20
    held = ga.device_info()["workspace_bytes"]
    table = 23 * (1 << 19) * 192 + 256               # 20-bit digits: 23 windows of 2^19 entries behind the header
    assert table < held <= 10 << 30, held
    assert held - table <= int(7.5 * 2**30), held - table
    ga.release_memory()


def test_release_memory_gives_everything_back_and_it_returns_on_demand(ga, O, width):
    """goldilocks_amd_release_memory: workspace, staging and the window table go (device_info counts them) and the next
    calls bring them back with the same results."""
    width(16)
    sigs, pks, msgs = _gen.signatures(O, 64, msglen=20, seed=b"bt/release", nkeys=4)
    reps = 80                                                # 5 120 signatures: the lane kernels and their workspace
    big = (np.concatenate([sigs] * reps), np.concatenate([pks] * reps), list(msgs) * reps)
    big[0][7, 3] ^= 1
    first = np.asarray(ga.ed448_verify_batch(*big))
    assert (first == -1).sum() == len(first) - 1 and first[7] == 0
    held = ga.device_info()["workspace_bytes"]
    table = 28 * 32768 * 192 + 256                           # 16-bit digits: 28 windows of 2^15 entries behind the header
    assert ga.get_base_table_bits() == 16 and held > table
    ga.release_memory(ga.RELEASE_BASE_TABLE)
    assert ga.get_base_table_bits() == 0 and ga.device_info()["workspace_bytes"] == held - table
    ga.release_memory()
    assert ga.device_info()["workspace_bytes"] == 0
    assert (np.asarray(ga.ed448_verify_batch(*big)) == first).all()
    assert ga.get_base_table_bits() == 16 and table < ga.device_info()["workspace_bytes"] <= held   # (what THIS batch needs)
    with pytest.raises(ga.GoldilocksAmdError):
        ga.release_memory(8)
