#!/usr/bin/env python3
"""bench.py -- Ed448 variable-base scalarmuls/s, batch 2^20 per GPU (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` (RANK/LOCAL_RANK/WORLD_SIZE in the environment), or plainly as `python bench.py --gpus
N`: then this process starts N fresh child processes, one per rank, before it has touched a GPU
(libgoldilocks_amd/shard.py: it never imports torch itself) and forwards rank 0's JSON line.  Ranks map
to the visible devices modulo their count; RCCL carries the barrier/MAX when every rank has its own
GPU, gloo when ranks share one (a 2-rank run on a 1-GPU box).

One "step" = one pass of the hot path over a batch that is already resident in HBM, i.e. one kernel
launch through the C ABI (goldilocks_amd_*_dev).  Ranks own independent batches (weak scaling, no
collective on the data path), or contiguous slices of one global batch with --global-log2-batch
(BASELINE config 5:  --workload verify --global-log2-batch 24 --gpus 8  -> 2^21 per GPU).
Rank 0 prints ONE JSON line.

Objects on the line besides the driver's contract:
  per_gpu       each rank's own throughput, device and average kernel time (config 5 asks for it)
  roofline      HBM roofline of the timed kernel: algorithmic bytes per op x ops per
                launch / average launch duration from HIP events on the launch stream, against 8 TB/s.
                The path is integer-multiply bound, so this fraction is tiny by construction;
                "mac" carries the honest ceiling: 32x32->64 MAC/s against the measured
                v_mad_u64_u32 peak (profiles/r01/ubench.txt).
  configs       (default N = 1 run only) the other BASELINE configs timed for a few steps each in
                the same process: fixed-base comb in LDS (config 3), the built-in base point, verify
                (config 4), and the opt-in digit-addressed tables for public scalars (*_fast).
  end_to_end    (default N = 1 run only) the host-array entry points (*_batch: pageable host arrays in,
                host arrays out, H2D + kernels + D2H) for the three headline paths, library defaults
                (SURVEY 8d "report both device-resident and end-to-end"; never `value`).
  cpu_baseline  the REAL reference (arch_x86_64 path, oracle/_ref; two builds: generic x86-64 and
                -march=x86-64-v3, the flags the reference's own Makefile.custom:63 -march=native gives on
                a current host) -- or the oracle port if those did not travel -- on the host cores: a
                single-thread figure by the reference's own benchmark method
                (test/bench_goldilocks.cxx:73-143: 50 samples x 20 iterations, 2 + 2 trimmed, mean; its
                :190 "Point scalarmul" line) and a short thread sweep over a bounded sample of the same
                batch.  This leg is also where the oracle checks 256 lanes of every config's output.

The table access of every timed call is passed PER CALL (the *_ex entry points, flags =
GOLDILOCKS_AMD_CALL_TABLES_*); the process-wide default of the library is never touched.  The headline
runs in the library's default mode, index-independent (the reference's constant-time contract).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LOG2_BATCH = 20
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
# The integer-multiply ceiling: the best issue rate tools/ubench measured for ANY quarter-rate VALU instruction on this
# part, 600 G wave-instructions/s (v_mul_u32_u24 599.8, v_mul_hi_u32 596; profiles/r01/ubench.txt, 8 waves per SIMD) x 64
# lanes.  The multiply-accumulates themselves top out lower (v_mad_i64_i32 589 G = 37.7 T, v_mad_u64_u32 579 G = 37.0 T); the
# paper figure is 1024 SIMDs x 16 lanes x 2.4 GHz = 39.3 T.  All three are on the line (roofline.mac.peaks).
# The ceiling every kernel here actually runs against: a SIMD issues ONE VALU instruction per four cycles whatever its kind
# (profiles/r01/ubench.txt: 4.1 - 4.4 cycles for every multiply / 64-bit / 3-operand instruction): 1024 SIMDs x 2.4 GHz / 4.
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4          # 614.4 G wave-instructions/s at the nominal clock (600 G measured, tools/ubench)
VALU_MAC_PEAK = 38.4e12
VALU_MAC_PEAKS = {"used": "best_issue_rate_measured", "best_issue_rate_measured": 38.4, "v_mad_i64_i32_measured": 37.7,
                  "v_mad_u64_u32_measured": 37.0, "paper_quarter_rate_at_2.4GHz": 39.3, "unit": "T MAC/s",
                  "source": "profiles/r01/ubench.txt (tools/ubench.hip, wall clock)"}

# Per-workload figures.  bytes: algorithmic I/O per op.  macs: 32x32->64 multiply-
# accumulates per op, counted by the host checker build of the same lane code
# (tests/test_hostsim.py::test_mac_counts_match_bench keeps these in step with the code).
WORKLOADS = {
    # macs: the reference's 5-bit window algorithm (2175 M x 192 + 1785 S x 136 + 17 mulw x 16), which the library ran for
    # public scalars until round 6; macs_index_independent: what it runs in both table-access modes (csrc/montgomery.hpp): 446 ladder steps of 5M + 4S + mulw, the y-recovery,
    # the shared-inversion chain (3 M) and an eighth of an inversion (8 operations per lane at 2^20)
    "varbase": dict(metric="Ed448 variable-base scalarmuls/sec", unit="scalarmuls/s", bytes=568, macs=660_632,
                    macs_index_independent=681_344 + 576 + 63_616 // 8, macs_ladder=681_344, macs_inversion=63_616,
                    desc="goldilocks_448_point_scalarmul, variable base, random scalars"),
    # a caller's precomputed_s: from 2^18 operations on the table is re-combed to 4 x 7 x 16 per call (k_import_comb +
    # k_recomb_big, 0.35 ms, inside the timed step) and multiplied by k_base_scalarmul_ct; below, the 5 x 5 x 18 comb
    "fixed": dict(metric="Ed448 fixed-base scalarmuls/sec", unit="scalarmuls/s", bytes=312, macs=101_664,
                  macs_reference_comb=138_848, recomb_min=1 << 18,
                  desc="goldilocks_448_precomputed_scalarmul, a caller's 5x5x18 comb table: re-combed to 4x7x16 per call, staged in LDS"),
    "base": dict(metric="Ed448 base-point scalarmuls/sec", unit="scalarmuls/s", bytes=312, macs=36_480, base_table_additions=True,
                 macs_index_independent=101_664,   # the library's 4 x 7 x 16 comb of the base point, staged in LDS
                 desc="goldilocks_448_precomputed_scalarmul(precomputed_base), the base point's window table"),
    # half-size scalars (csrc/lattice.hpp): two decodings + two window tables + one 45-window ladder over both
    # points + two correcting additions + 28 base-point additions
    # macs_own_key: every lane decodes its key and builds the key's table itself (what "verify_distinct" runs);
    # macs_shared_keys: the key has a pooled window table (one decoding and one table per DISTINCT key of the batch,
    # kernels_verify.hip); macs_key_comb: the key has a fixed-base comb of its own (keys that sign at least 16 of the
    # batch's signatures on average): no ladder, no decoding of R -- the key's 4 x 7 x 16 comb, the base point's
    # additions, R's test with a shared inversion; per key 3.46 M more (its decoding, 432 doublings, 256 entries);
    # macs_key_comb_wide: keys with 256 signatures or more get 4 x 8 x 14 combs: 13 doublings + 55 additions, 7.09 M per
    # key (512 entries) = 6 922 per signature at 2^10 keys of 2^20 signatures; macs_key_comb_xwide: from 1 024 signatures
    # per key on 5 x 9 x 10 combs -- what 2^20 signatures of 2^10 keys run at: 9 doublings + 49 additions, 18.77 M per
    # key as the checker counts it (1 280 entries; the device shares an inversion between 32 of them) = 18 325 per signature
    "verify": dict(metric="Ed448 verifies/sec", unit="verifies/s", bytes=207, macs=124_440 + 18_325, base_table_additions=True,
                   macs_key_comb=149_976, macs_per_key_comb=3_458_392,
                   macs_key_comb_wide=136_984, macs_per_key_comb_wide=7_088_472,
                   macs_key_comb_xwide=124_440, macs_per_key_comb_xwide=18_765_352,
                   macs_pooled_tables=522_128 + 87, macs_own_key=611_480, macs_shared_keys=522_128, keys=1024,
                   desc="goldilocks_ed448_verify, 32-byte messages, 2^10 distinct keys (SURVEY 8d), 1% corrupted"),
    "verify_distinct": dict(metric="Ed448 verifies/sec, every signature under its own key", unit="verifies/s", bytes=207,
                            macs=611_480, keys=None, base_table_additions=True,
                            desc="goldilocks_ed448_verify, 32-byte messages, as many distinct keys as signatures, 1% corrupted"),
    "sign": dict(metric="Ed448 signatures/sec", unit="signatures/s", bytes=260, macs=None,
                 desc="goldilocks_ed448_sign, 32-byte messages, no context"),
    "x448": dict(metric="X448 shared secrets/sec", unit="shared secrets/s", bytes=172, macs=None,
                 desc="goldilocks_x448, random peer public keys"),
    "direct": dict(metric="Ed448 wire-format scalarmuls/sec", unit="scalarmuls/s", bytes=172, macs=None,
                   desc="goldilocks_448_direct_scalarmul, 56-byte encodings in and out"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-batch", type=int, default=LOG2_BATCH, help="operations per GPU (weak scaling)")
    ap.add_argument("--global-log2-batch", type=int, default=None,
                    help="one global batch of 2^G operations cut into contiguous per-rank slices (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE configs on the default run")
    ap.add_argument("--workload", default="varbase", choices=sorted(WORKLOADS))
    ap.add_argument("--table-access", default="index-independent", choices=["fast", "index-independent"],
                    help="GOLDILOCKS_AMD_CALL_TABLES_* of every timed call: how tables are read for (possibly secret) "
                         "scalars; index-independent is the library's default and the reference's contract, fast the "
                         "opt-in for public scalars")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-array (PCIe-inclusive) legs")
    ap.add_argument("--stub-step-ms", type=float, default=None,
                    help="launcher/timing self-test without a GPU: a step sleeps this long (tests only)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------- workloads

class Ctx(object):
    """What a workload needs: the binding, torch, the rank's batch size and launch stream."""

    def __init__(self, ga, np, torch, n, rank):
        self.ga, self.np, self.torch, self.n, self.rank = ga, np, torch, n, rank
        self.stream = torch.cuda.current_stream().cuda_stream
        self._pairs = None

    def pairs(self):
        """Synthetic (point, scalar) batch from the SHAKE256 stream "bench_varbase_v1/<rank>/..."
        (tests/_gen.py): scalars uniform below 2^446; base points = k*B for stream scalars k (all
        distinct), computed on the device by the fixed-base kernel.  Rank 0's 2^20 batch is the one
        whose outputs are pinned by tests/golden/f6_bench_digest.json (made with the real reference)."""
        if self._pairs is None:
            import _gen
            torch, np, n = self.torch, self.np, self.n

            def stream_scalars(what):
                s = _gen.stream_scalars(n, b"bench_varbase_v1/%d/%s" % (self.rank, what))
                return torch.from_numpy(s.view(np.int64)).cuda()

            scalars, k = stream_scalars(b"scalar"), stream_scalars(b"base")
            bases = torch.empty((n, 32), dtype=torch.int64, device="cuda")
            self.ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), n, None)
            torch.cuda.synchronize()
            self._pairs = (bases, scalars)
        return self._pairs


SAMPLE = 256   # lanes of every config the oracle re-computes in the cpu_baseline leg


def make_workload(name, cx, access):
    """-> dict(step, kernel, check() -> (ok, text, extra), sample() -> what the oracle needs to re-compute
    the first SAMPLE lanes (host arrays only; the oracle itself is touched in cpu_baseline_leg alone)).
    access: "index-independent" | "fast" -- passed to every timed call as its GOLDILOCKS_AMD_CALL_TABLES_* flags."""
    ga, np, torch, n, stream = cx.ga, cx.np, cx.torch, cx.n, cx.stream
    ct = access == "index-independent"
    flags = ga.CALL_TABLES_INDEX_INDEPENDENT if ct else ga.CALL_TABLES_FAST
    host = lambda t: t.cpu().numpy()
    if name == "varbase":
        bases, scalars = cx.pairs()
        out = torch.empty_like(bases)
        step = lambda: ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), scalars.data_ptr(), n, stream, flags=flags)

        def check():
            stv = torch.empty(n, dtype=torch.int32, device="cuda")
            ga.dev("point_pred", stv.data_ptr(), out.data_ptr(), None, 1, n, None)
            ok = int((stv == -1).sum()) == n
            text, extra = "every output is a valid point", {}
            dig_path = os.path.join(ROOT, "tests", "golden", "f6_bench_digest.json")
            known = json.load(open(dig_path))["digest_shake256_32"] if os.path.exists(dig_path) else {}
            log2 = n.bit_length() - 1
            if cx.rank == 0 and (1 << log2) == n and str(log2) in known:
                import hashlib
                ser = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
                ga.dev("point_encode", ser.data_ptr(), out.data_ptr(), n, None)
                digest = hashlib.shake_256(ser.cpu().numpy().tobytes()).hexdigest(32)
                match = digest == known[str(log2)]
                ok = ok and match
                text += "; SHAKE256 digest of all outputs equals the reference's (golden F6)"
                extra["full_batch_digest"] = {"shake256_32": digest, "matches_reference_fixture": match}
            return ok, text, extra

        def sample():
            m = min(n, SAMPLE)
            return dict(kind="varbase", bases=host(bases[:m]).view(np.uint64), scalars=host(scalars[:m]).view(np.uint64),
                        got=ga.point_encode_batch(host(out[:m]).view(np.uint64)))
        return dict(step=step, kernel="k_point_scalarmul_ct", check=check, sample=sample)   # (the ladder in both table-access modes since round 6)
    if name in ("fixed", "base"):
        _, scalars = cx.pairs()
        out = torch.empty((n, 32), dtype=torch.int64, device="cuda")
        if name == "fixed":    # BASELINE config 3: a caller's precomputed_s -> a comb staged in LDS (either mode)
            tab = torch.from_numpy(ga.precomputed_base().view(np.int64)).cuda()
            step = lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), tab.data_ptr(), scalars.data_ptr(), n, stream,
                                  flags=flags)
            kernel = "k_base_scalarmul_ct" if n >= WORKLOADS["fixed"]["recomb_min"] else "k_precomputed_scalarmul"
        else:                  # the built-in base point: its LDS comb (index-independent) or its 16-bit window table (fast)
            tab = None
            step = lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), None, scalars.data_ptr(), n, stream, flags=flags)
            kernel = "k_base_scalarmul_ct" if ct else "k_base_scalarmul"

        def check():
            # s*B once more through a kernel that is not timed anywhere in this run (so the rocprofv3 averages
            # of the timed kernels stay clean) and shares no table with them: the two-point ladder s*P + 0*P
            # with the generator handed over as an ordinary caller's point
            m = min(n, 1 << 14)
            base_pt = torch.from_numpy(np.repeat(ga.point_base().reshape(1, 32), m, axis=0).view(np.int64)).cuda()
            zero = torch.zeros((m, 7), dtype=torch.int64, device="cuda")
            alt = torch.empty((m, 32), dtype=torch.int64, device="cuda")
            ga.dev("point_double_scalarmul", alt.data_ptr(), base_pt.data_ptr(), scalars.data_ptr(), base_pt.data_ptr(),
                   zero.data_ptr(), m, None, flags=ga.CALL_TABLES_FAST)
            st = torch.empty(m, dtype=torch.int32, device="cuda")
            ga.dev("point_pred", st.data_ptr(), alt.data_ptr(), out.data_ptr(), 0, m, None)
            return int((st == -1).sum()) == m, "first %d results equal s*P + 0*P from the two-point window ladder, P = the generator as a caller's point" % m, {}

        def sample():
            m = min(n, SAMPLE)
            return dict(kind="fixed", scalars=host(scalars[:m]).view(np.uint64),
                        got=ga.point_encode_batch(host(out[:m]).view(np.uint64)))
        return dict(step=step, kernel=kernel, check=check, sample=sample, keep=[tab])
    if name == "direct":       # wire format in and out: 56-byte encodings, decode + ladder + encode fused
        bases, scalars = cx.pairs()
        enc_in = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        ga.dev("point_encode", enc_in.data_ptr(), bases.data_ptr(), n, None)
        enc_out = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        st_direct = torch.empty(n, dtype=torch.int32, device="cuda")
        step = lambda: ga.dev("direct_scalarmul", enc_out.data_ptr(), st_direct.data_ptr(), enc_in.data_ptr(),
                              scalars.data_ptr(), 0, 0, n, stream, flags=flags)

        def check():
            ref = torch.empty_like(bases)
            ga.dev("point_scalarmul", ref.data_ptr(), bases.data_ptr(), scalars.data_ptr(), n, None, flags=flags)
            ref_enc = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
            ga.dev("point_encode", ref_enc.data_ptr(), ref.data_ptr(), n, None)
            ok = bool((ref_enc == enc_out).all()) and int((st_direct == -1).sum()) == n
            return ok, "every output equals encode(point_scalarmul(decode(input))) computed by the separate kernels", {}
        return dict(step=step, kernel="k_direct_scalarmul_ct", check=check)
    import _gen
    if name in ("sign", "x448"):
        nb = 57 if name == "sign" else 56
        sk = torch.from_numpy(np.frombuffer(_gen.stream(b"bench_%s_v1/%d/sk" % (name.encode(), cx.rank), nb * n),
                                            np.uint8).reshape(n, nb).copy()).cuda()
        if name == "sign":
            pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
            ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), n, None)
            msg = torch.from_numpy(np.frombuffer(_gen.stream(b"bench_sign_v1/%d/msg" % cx.rank, 32 * n), np.uint8)
                                   .reshape(n, 32).copy()).cuda()
            sig_out = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
            step = lambda: ga.dev("ed448_sign", sig_out.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None,
                                  32, 0, None, 0, n, stream, flags=flags)

            def check():
                st = torch.empty(n, dtype=torch.int32, device="cuda")
                ga.dev("ed448_verify", st.data_ptr(), sig_out.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0,
                       None, 0, n, None)
                return int((st == -1).sum()) == n, "every signature verifies (ed448_verify kernel)", {}
            return dict(step=step, kernel="k_ed448_sign_ct" if ct else "k_ed448_sign", check=check)
        pub = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        ga.dev("x448", pub.data_ptr(), None, None, sk.data_ptr(), n, None)
        peer = pub.view(n // 2, 2, 56).flip(1).reshape(n, 56).contiguous()      # lane i meets lane i^1's public key
        shared = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        st448 = torch.empty(n, dtype=torch.int32, device="cuda")
        step = lambda: ga.dev("x448", shared.data_ptr(), st448.data_ptr(), peer.data_ptr(), sk.data_ptr(), n, stream)

        def check():
            pairs = shared.view(n // 2, 2, 56)
            ok = bool((pairs[:, 0] == pairs[:, 1]).all()) and int((st448 == -1).sum()) == n
            return ok, "Diffie-Hellman symmetry: X448(a, pub_b) == X448(b, pub_a) for every neighbour pair", {}
        return dict(step=step, kernel="k_x448", check=check)
    # verify: signatures over 32-byte messages from 1024 distinct keys, produced by the
    # library's own derive/sign kernels (bit-exact vs the reference: tests/test_gpu_parity.py); 1 % corrupted
    v = verify_inputs(cx, distinct=name == "verify_distinct")
    d_sig, d_pk, d_msg = (torch.from_numpy(v[k]).cuda() for k in ("sig", "pk", "msg"))
    status = torch.empty(n, dtype=torch.int32, device="cuda")
    step = lambda: ga.dev("ed448_verify", status.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(),
                          None, 32, 0, None, 0, n, stream)

    def check():
        got = (status == -1).cpu().numpy()
        return bool((got == ~v["bad"]).all()), "accepted lanes == uncorrupted lanes (signatures made by the sign kernel)", {}

    def sample():
        m = min(n, SAMPLE)
        return dict(kind="verify", sigs=v["sig"][:m], pks=v["pk"][:m], msgs=[x.tobytes() for x in v["msg"][:m]],
                    got=host(status[:m]))
    # 2^10 keys: every key gets a fixed-base comb, the wide one at 2^10 signatures per key (kernels_verify.hip);
    # all-distinct keys: every lane for itself
    # (the key-comb verification is a first pass, S*B computed ahead beside the combs' build, and a finish kernel: the
    # step's memory traffic is the three kernels' together; `kernel` names the dominant one)
    if name == "verify_distinct":
        return dict(step=step, kernel="k_ed448_verify", check=check, sample=sample)
    # ... and which comb the batch's keys got decides the first pass's kernel: asked of the library after the steps)
    def kernels_after():
        teeth = ga.last_verify_key_counts(teeth=True)[3]
        main = {7: "k_ed448_verify_keycomb", 8: "k_ed448_verify_keycomb_wide", 9: "k_ed448_verify_keycomb_xwide"}.get(teeth)
        if main is None:
            return "k_ed448_verify", None, None
        tag = {7: "", 8: "_wide", 9: "_xwide"}[teeth]
        V = WORKLOADS["verify"]
        macs = V["macs_key_comb" + tag] + V["macs_per_key_comb" + tag] * V["keys"] // n
        return main, (main,) + VERIFY_STEP_KERNELS, macs
    return dict(step=step, kernel="k_ed448_verify_keycomb_xwide", check=check, sample=sample, kernels_after=kernels_after,
                traffic_kernels=("k_ed448_verify_keycomb_xwide",) + VERIFY_STEP_KERNELS)


# the kernels of a key-comb verification step besides its main one (their traffic and instructions are summed into the
# step's: until round 6 the keys' preparation -- teeth, comb entries, the hash set -- was left out of the sum; the main
# kernel of a device-resident batch finishes its own positions, k_ed448_verify_keycomb_finish is the host-array pipeline's)
VERIFY_STEP_KERNELS = ("k_verify_base_part", "k_verify_key_teeth", "k_verify_key_combs", "k_verify_dedupe")


def verify_inputs(cx, distinct=False):
    """BASELINE config 4's synthetic input (host arrays): n signatures over 32-byte messages from 1024 keys
    (SURVEY 8d), 1 % corrupted (and lane 5 always, so that the oracle's sample of the first lanes holds a reject).
    distinct: every signature under a key of its own instead (what a verifier sees when no key repeats)."""
    attr = "_verify_distinct" if distinct else "_verify"
    if getattr(cx, attr, None) is None:
        import _gen
        ga, np, n = cx.ga, cx.np, cx.n
        bad = np.random.default_rng(cx.rank + 99).random(n) < 0.01
        if n > 5:
            bad[5] = True
        # every signature is a signature of its own (its own message, hence its own R, challenge and S): signed on the
        # device by the library's sign kernel.  (Until round 4 the 2^10-key batch was 4 096 distinct signatures drawn
        # 2^20 times: the same S and the same challenge 256 times over, i.e. table entries that the caches had already
        # seen -- kinder to the gathers than a real batch.)
        torch = cx.torch
        tag = b"distinct" if distinct else b"1024keys"
        nk = n if distinct else 1024
        sk_k = np.frombuffer(_gen.stream(b"bench_verify_v2/%d/sk-%s" % (cx.rank, tag), 57 * nk), np.uint8).reshape(nk, 57)
        key_of = np.arange(n) if distinct else np.random.default_rng(cx.rank).integers(0, nk, n)
        sk = torch.from_numpy(np.ascontiguousarray(sk_k[key_of])).cuda()
        msg = torch.from_numpy(np.frombuffer(_gen.stream(b"bench_verify_v2/%d/msg-%s" % (cx.rank, tag), 32 * n), np.uint8)
                               .reshape(n, 32).copy()).cuda()
        pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
        sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
        ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), n, None)
        ga.dev("ed448_sign", sig.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
        torch.cuda.synchronize()
        sig_h, pk_h, msg_h = sig.cpu().numpy(), pk.cpu().numpy(), msg.cpu().numpy()
        del sk, msg, pk, sig
        sig_h = np.ascontiguousarray(sig_h)
        sig_h[bad, 5] ^= 0x20
        setattr(cx, attr, dict(sig=sig_h, pk=np.ascontiguousarray(pk_h), msg=np.ascontiguousarray(msg_h), bad=bad))
    return getattr(cx, attr)


# ---------------------------------------------------------------------------------------- end to end

def end_to_end(cx, reps=5):
    """The host-array entry points (goldilocks_448_point_scalarmul_batch, _precomputed_scalarmul_batch,
    goldilocks_ed448_verify_batch): pageable host arrays in, host arrays out -- H2D, kernels and D2H inside
    the timed call, library defaults (index-independent tables).  One untimed call first (it sizes the
    library's staging pool), then the median of `reps` (every call's time is on the line too: a call whose
    transfers do not keep the GPU busy finds its clocks down, and the rate of such a call can be a third lower).  Output arrays are allocated and touched beforehand:
    what is timed is the library, not the page faults of a fresh numpy array."""
    ga, np, n = cx.ga, cx.np, cx.n
    L = ga.lib()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    bases, scalars = cx.pairs()
    b_h, s_h = bases.cpu().numpy().view(np.uint64), scalars.cpu().numpy().view(np.uint64)
    out = np.zeros((n, 32), dtype=np.uint64)
    v = verify_inputs(cx)
    msg = v["msg"]
    mptr = (msg.ctypes.data + 32 * np.arange(n, dtype=np.uint64)).astype(np.uint64)   # const uint8_t *message[n]
    mlen = np.full(n, 32, dtype=np.uint64)                                           # size_t message_len[n]
    st = np.zeros(n, dtype=np.int32)
    tab = C.c_void_p.in_dll(L, "goldilocks_448_precomputed_base")
    calls = {
        "varbase": (lambda: L.goldilocks_448_point_scalarmul_batch(ptr(out), ptr(b_h), ptr(s_h), n), 568,
                    "goldilocks_448_point_scalarmul_batch"),
        "fixed": (lambda: L.goldilocks_448_precomputed_scalarmul_batch(ptr(out), tab, ptr(s_h), n), 312,
                  "goldilocks_448_precomputed_scalarmul_batch(goldilocks_448_precomputed_base)"),
        "verify": (lambda: L.goldilocks_ed448_verify_batch(ptr(st), ptr(v["sig"]), ptr(v["pk"]), ptr(mptr), ptr(mlen), 0,
                                                           None, 0, n), 207, "goldilocks_ed448_verify_batch"),
    }
    # the link, measured here and now: 256 MiB from / to pageable host memory, the kind of memory the calls below get
    # (best of three; tools/probes/h2d_probe.py has the pinned figures beside them)
    torch = cx.torch
    probe_h = torch.from_numpy(np.ones(1 << 28, dtype=np.uint8))
    probe_d = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    link = {}
    for direction, (dst, src) in (("h2d", (probe_d, probe_h)), ("d2h", (probe_h, probe_d))):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(src)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        link[direction] = (1 << 28) / best / 1e9
    del probe_h, probe_d
    io_bytes = {"varbase": (312, 256), "fixed": (56, 256), "verify": (203 + 8, 4)}   # (in, out) per operation; verify: + the 8-byte offset
    res = {"link_gbs": {k: round(v, 1) for k, v in link.items()}}
    for name, (call, nbytes, what) in calls.items():
        if call():
            raise RuntimeError(L.goldilocks_amd_last_error().decode())
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            rc = call()
            times.append(time.perf_counter() - t0)
            if rc:
                raise RuntimeError(L.goldilocks_amd_last_error().decode())
        t = sorted(times)[len(times) // 2]
        b_in, b_out = io_bytes[name]
        # the link is full duplex: the busier direction bounds the call.  pcie_frac near 1: the call is PCIe-bound
        # and no kernel can help it; well below: the device (or the host's packing) is the bound.
        frac = max(b_in * n / t / 1e9 / link["h2d"], b_out * n / t / 1e9 / link["d2h"])
        res[name] = {"value": n / t, "unit": WORKLOADS[name]["unit"], "ms_per_call": t * 1e3, "entry_point": what,
                     "pcie_bytes_per_op": nbytes, "pcie_bytes_in_out": [b_in, b_out], "pcie_frac": frac,
                     "host_memory": "pageable", "reps": reps, "ms_every_call": [round(x * 1e3, 2) for x in times]}
    ok = bool(((st == -1) == ~v["bad"]).all())
    res["verify"]["check"] = "accepted lanes == uncorrupted lanes" if ok else "FAILED"
    return res, ok


# ---------------------------------------------------------------------------------------- CPU baseline

def host_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(math.ceil(quota))))
    return usable, affinity, quota


def cpu_info():
    model, flags = "unknown", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            if line.startswith("flags") and not flags:
                flags = set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return model, flags


def reference_tool_rows(build_label):
    """Runs oracle/_ref/bench_goldilocks_<build> --micro (the reference's test/bench_goldilocks.cxx, :73-143 its Benchmark
    class, :190 'Point scalarmul') and returns its Ed448-Goldilocks rows in seconds per operation; None if the binary is
    not there (it is built where /root/reference is)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_goldilocks_" + ("x86_64_v3" if build_label == "x86_64_v3" else "x86_64"))
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--micro"], capture_output=True, text=True, timeout=120).stdout
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)}
    scale = {"ns": 1e-9, "µs": 1e-6, "us": 1e-6, "ms": 1e-3, "s": 1.0}
    rows, section = {}, ""
    for line in out.splitlines():
        if line.startswith(("Micro-benchmarks for", "Macro-benchmarks for")):
            section = line.split("for", 1)[1].strip(" :")
        m = re.match(r"^(.*?):\s+([0-9.]+)\s*(ns|µs|us|ms|s)\s", line)
        if m and "448" in section:
            rows[m.group(1).strip()] = float(m.group(2)) * scale[m.group(3)]
    wanted = ("Point scalarmul", "Point precmp scalarmul", "Point double scalarmul", "Point double scalarmul_v",
              "EdDSA sign", "EdDSA verify", "RFC 7748 shared secret")
    sel = {k: rows[k] for k in wanted if k in rows}
    return {"tool": "test/bench_goldilocks.cxx --micro (compiled by oracle/Makefile against the reference's arch_x86_64 path, " + build_label + ")",
            "seconds_per_op": sel,
            "point_scalarmul_per_s": 1.0 / sel["Point scalarmul"] if sel.get("Point scalarmul") else None, "threads": 1}


def cpu_baseline_leg(np, bases_h, scalars_h, samples, budget_s=9.0):
    """Time the reference's CPU path on the host cores and let the oracle re-compute SAMPLE lanes of every
    config.  The only place in this file that touches oracle/ (test infrastructure).
    samples: {config name: workload sample()}.  -> (cpu_baseline object, {config name: (ok, text)})."""
    from _libs import oracle, REF_X86_SO, REF_X86_V3_SO
    import _gen
    O = oracle()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    model, cpuflags = cpu_info()
    builds = []   # (label, what, function pointer or None)
    for so, label, what, need in (
            (REF_X86_SO, "x86_64_generic", "reference arch_x86_64 path, gcc -O2 -mtune=generic (oracle/Makefile)", set()),
            (REF_X86_V3_SO, "x86_64_v3", "reference arch_x86_64 path, gcc -O2 -march=x86-64-v3: BMI2 mulx + AVX2, what "
             "the reference's Makefile.custom:63 -march=native gives on a current host", {"avx2", "bmi2"})):
        if os.path.exists(so) and need <= cpuflags:
            try:
                builds.append((label, what, C.cast(C.CDLL(so).goldilocks_448_point_scalarmul, C.c_void_p)))
            except OSError:
                pass
    kind = "reference" if builds else "port"
    if not builds:
        builds.append(("oracle_port", "oracle/gold_oracle.c (ref64-shaped restatement, gcc -O3)", None))
    usable, affinity, quota = host_cores()
    n = len(scalars_h)

    # (1) one thread, the reference's Benchmark method: 50 samples x 20 calls, drop 2 + 2, mean -- every build
    nsamples, ntests, discard = 50, 20, 2
    m1 = min(n, nsamples * ntests)
    b1, s1 = np.ascontiguousarray(bases_h[:m1]), np.ascontiguousarray(scalars_h[:m1])
    singles = {}
    for label, what, fn in builds:
        times = np.zeros(nsamples, dtype=np.float64)
        O.orc_bench_extern_scalarmul(fn, p(b1), p(s1), m1, 3, ntests, p(times))          # warm tables and caches
        O.orc_bench_extern_scalarmul(fn, p(b1), p(s1), m1, nsamples, ntests, p(times))
        us_per_op = float(np.sort(times)[discard:nsamples - discard].mean()) / ntests * 1e6
        singles[label] = {"us_per_op": us_per_op, "value": 1e6 / us_per_op, "build": what}
    best_label = max(singles, key=lambda k: singles[k]["value"])
    label, what, fn = next(b for b in builds if b[0] == best_label)
    single = dict(singles[best_label], method="50 samples x 20 iterations, 2 low + 2 high dropped, mean "
                                              "(test/bench_goldilocks.cxx:73-143, :190)")

    # (2) thread sweep of the faster build over a bounded sample of the same batch; every point about budget/4 seconds
    per_point = max(0.5, (budget_s - 1.0) / 3.0)
    sweep = []
    for t in sorted({1, max(1, usable // 2), usable}):
        t = min(t, 256)                                    # the harness has 256 thread slots
        m = int(min(n, max(t * 64, per_point * t * single["value"])))
        b, s = np.ascontiguousarray(bases_h[:m]), np.ascontiguousarray(scalars_h[:m])
        out = np.empty((m, 32), dtype=np.uint64)
        t0 = time.perf_counter()
        if fn:
            O.orc_extern_scalarmul_batch(fn, p(out), p(b), p(s), m, t)
        else:
            O.orc_point_scalarmul_batch(p(out), p(b), p(s), m, t)
        dt = time.perf_counter() - t0
        sweep.append({"threads": t, "ops": m, "seconds": dt, "value": m / dt})
    best = max(sweep, key=lambda r: r["value"])
    try:
        import subprocess
        gcc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:   # noqa
        gcc = "unknown"
    res = {"value": best["value"], "unit": "scalarmuls/s", "cores": best["threads"], "kind": kind,
           "sample": "best of a thread sweep %s over the first %d of this batch's (point, scalar) pairs; %s" %
                     ([r["threads"] for r in sweep], best["ops"], what),
           "build": best_label, "single_thread": single, "single_thread_by_build": singles, "sweep": sweep,
           "compiler": gcc + " (in the build container)", "cpu_model": model, "affinity_cores": affinity,
           "cgroup_quota_cores": quota, "usable_cores": usable}

    # (2b) the reference's OWN tool: test/bench_goldilocks.cxx --micro, compiled in the build container against the same
    # reference library (oracle/Makefile), run here on the host as it is -- its rows, as it prints them
    res["reference_tool"] = reference_tool_rows(best_label)

    # (3) the oracle re-computes the first lanes of every config
    checks = {}
    for cname, smp in samples.items():
        if smp["kind"] == "varbase":
            want = _gen.oracle_encode(_gen.oracle_varbase(O, smp["bases"], smp["scalars"]))
            what_chk = "oracle's goldilocks_448_point_scalarmul (src/goldilocks.c:405-465)"
        elif smp["kind"] == "fixed":
            want = _gen.oracle_encode(_gen.oracle_fixed(O, smp["scalars"]))
            what_chk = "oracle's goldilocks_448_precomputed_scalarmul (src/goldilocks.c:830-877)"
        else:
            want = _gen.oracle_verify(O, smp["sigs"], smp["pks"], smp["msgs"])
            what_chk = "oracle's goldilocks_ed448_verify (src/eddsa.c:253-306), %d rejects among them" % int((want == 0).sum())
        same = bool((np.asarray(smp["got"]) == want).all())
        checks[cname] = (same, "first %d lanes equal the %s" % (len(want), what_chk))
    return res, checks


# ---------------------------------------------------------------------------------------- the clock the chip holds

class ClockSampler:
    """The shader clock of one device while a timed region runs: /sys/class/drm/card*/device/pp_dpm_sclk (the level marked
    '*'), read every 25 ms on a thread.  The VALU-issue ceiling is quoted against the nominal 2.4 GHz AND against what the
    part sustained under this very load (it is power-limited: 2.30 - 2.32 GHz on the headline, profiles/r05/clock_under_load.txt)."""

    def __init__(self, pci=None):
        import glob
        self.path, self.samples, self._stop, self._thread = None, [], False, None
        # (a box shows every GPU of its host in sysfs, whichever one the process was given: the device is found by its PCI address)
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            real = os.path.realpath(os.path.dirname(f))
            if pci is None or pci.lower() in real.lower():
                self.path = f
                break

    def _read(self):
        try:
            for l in open(self.path):
                if "*" in l:
                    return float(l.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def __enter__(self):
        if self.path:
            import threading

            def run():
                while not self._stop:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.025)
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread:
            self._thread.join()

    def sustained_mhz(self):
        """median of the samples taken under load (idle levels -- below 1 GHz -- dropped); None if the file is not there"""
        v = sorted(x for x in self.samples if x >= 1000)
        return v[len(v) // 2] if v else None


# ---------------------------------------------------------------------------------------- one rank

def time_workload(torch, shard, w, steps, warmup, dist, backend, clock=None):
    """-> (seconds of this rank, MAX over ranks, per-step kernel ms from HIP events on the launch stream).
    clock: a ClockSampler that runs while the warm-up and the timed steps do."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if clock is None:
        clock = ClockSampler.__new__(ClockSampler)
        clock.path, clock.samples, clock._stop, clock._thread = None, [], False, None
    with clock:
        mine, worst = shard.timed_region(w["step"], steps, warmup, torch.cuda.synchronize, dist, backend,
                                         after_step=lambda i: ev[i + 1].record())
    return mine, worst, [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def kernel_source_sha16():
    """the kernels' sources of THIS tree (tools/summarize_prof.py stamps the same digest on the counters it condenses)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libgoldilocks_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, bench=None):
    """HBM bytes per launch (kernel: a name, or the names of the kernels that make up a step: their sum) from the PMC passes of tools/profile_round.sh (profiles/pmc_traffic.json) -- a measurement of
    another run, so it says which kernels it was taken on: the figure is only reported as `traffic` when the kernels'
    sources are still the ones it was measured on; otherwise traffic is null and the stale figure goes to traffic_stale."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    stamp = d.get("_measured_on") or {}
    current = stamp.get("kernel_source_sha16") == kernel_source_sha16()
    info = dict(stamp, current=current, file="profiles/pmc_traffic.json")
    # a kernel's traffic depends on the configuration it ran in (config 4 at the 20-bit and at the 24-bit base table gathers
    # from 2.2 and from 28.5 GiB): the passes are kept per bench run, and a figure is only reported for the run it was taken on
    if bench is not None and "_by_bench" in d:          # (a file from before round 6 has the one flat table only)
        d = d["_by_bench"].get(bench)
        if d is None:
            return None, dict(info, bench=bench, missing="no PMC pass for this configuration")
        info = dict(info, bench=bench)
    if isinstance(kernel, (tuple, list)):
        parts = [d.get(k) for k in kernel]
        return (sum(parts) if all(x is not None for x in parts) else None), dict(info, kernels=list(kernel))
    return d.get(kernel), info


def pmc_valu_insts(kernel, bench=None):
    """VALU wave-instructions per 2^20-operation launch of `kernel` (or of the kernels of a step: their sum) from the same
    stamped PMC passes (SQ_INSTS_VALU); None when they were not taken on the current kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    if (d.get("_measured_on") or {}).get("kernel_source_sha16") != kernel_source_sha16():
        return None
    v = d.get("_valu_insts") or {}
    if bench is not None and "_valu_by_bench" in d:
        v = d["_valu_by_bench"].get(bench) or {}
    if isinstance(kernel, (tuple, list)):
        parts = [v.get(k) for k in kernel]
        return sum(parts) if all(x is not None for x in parts) else None
    return v.get(kernel)


def base_table_windows(bits):
    """Digits of a scalar in the base point's window table of `bits`-bit digits (scalarmul.hpp bwt_windows)."""
    return -(-446 // bits)


def table_access_uses_base_table(name, table_access):
    """Verification multiplies the base point by public data whatever the mode; the base point's own multiplication
    takes its window table only with digit-addressed tables."""
    return name.startswith("verify") or table_access != "index-independent"


def pmc_bench_key(name, table_access, base_table_bits):
    """the run of tools/profile_round.sh whose counters belong to this configuration"""
    key = name + ("_fast" if table_access != "index-independent" and name in ("base", "sign") else "")
    if name.startswith("verify") and base_table_bits not in (0, 20):
        key += "%d" % base_table_bits
    return key


def roofline(name, kernel, n, avg_ms, table_access, traffic_kernels=None, base_table_bits=0, macs=None, sustained_mhz=None):
    spec = dict(WORKLOADS[name])
    bench_key = pmc_bench_key(name, table_access, base_table_bits)
    if macs:                # (verification: the multiply-accumulates of the comb geometry the batch's keys really got)
        spec["macs"] = macs
    # (one scalar times a variable base runs the table-free ladder in BOTH table-access modes since round 6)
    if (table_access == "index-independent" or name in ("varbase", "direct")) and spec.get("macs_index_independent"):
        spec["macs"] = spec["macs_index_independent"]
    elif spec.get("macs") and spec.get("base_table_additions") and base_table_bits:
        # the figures are priced for 16-bit digits (28 of them); the device's table may be wider: a mixed addition
        # (7 multiplications) less per digit saved
        spec["macs"] -= (28 - base_table_windows(base_table_bits)) * 7 * 192
    if n < spec.get("recomb_min", 0):
        spec["macs"] = spec["macs_reference_comb"]
    achieved = spec["bytes"] * n / (avg_ms * 1e-3) / 1e9
    traffic, measured_on = pmc_traffic(traffic_kernels or kernel, bench_key)
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic if measured_on and measured_on["current"] else None, "kernel": kernel, "kernel_ms_avg": avg_ms,
         "bytes_per_op": spec["bytes"],
         "note": "integer-multiply bound by construction (SURVEY 8d); the ceiling that matters is mac"}
    if measured_on:
        r["traffic_measured_on"] = measured_on
        if not measured_on["current"]:
            r["traffic_stale"] = traffic
    if spec.get("base_table_additions") and table_access_uses_base_table(name, table_access) and base_table_bits:
        # memory for additions: every digit of a scalar is one 192-byte entry of the base point's
        # table, i.e. two 128-byte lines, and with digits beyond 16 bits the table (28.5 GiB at 24) is in no cache
        r["traffic_of_base_table_gathers"] = base_table_windows(base_table_bits) * 256 * n
        r["traffic_note"] = ("most of `traffic` is the base point's window table, gathered on purpose: one entry per digit "
                             "instead of a mixed addition more per digit saved; the kernels do not wait for it")
    if spec["macs"]:
        macs = spec["macs"] * n / (avg_ms * 1e-3)
        r["mac"] = {"achieved": macs / 1e12, "peak": VALU_MAC_PEAK / 1e12, "unit": "T MAC/s",
                    "frac": macs / VALU_MAC_PEAK, "macs_per_op": spec["macs"], "peaks": VALU_MAC_PEAKS}
    # ... and the ceiling all of it runs against: VALU issue, one wave-instruction per SIMD per four cycles.  The instruction
    # count is SQ_INSTS_VALU of a 2^20-operation launch from the stamped PMC passes (scaled to this launch's n); the time
    # is this run's.  What is left below 1.0 is the clock the chip holds under this load (2.30 - 2.32 GHz at 1.31 kW:
    # power-limited, profiles/r05/clock_under_load.txt) and about five percent of stalls; the lever is instructions per operation.
    insts = pmc_valu_insts(traffic_kernels or kernel, bench_key)
    if insts:
        per_s = insts * (n / float(1 << LOG2_BATCH)) / (avg_ms * 1e-3)
        r["valu_issue"] = {"achieved": per_s / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
                           "frac": per_s / VALU_ISSUE_PEAK, "valu_instructions_per_op": insts * 64 / float(1 << LOG2_BATCH),
                           "source": "SQ_INSTS_VALU of run '%s', profiles/pmc_traffic.json (same kernel sources); peak = 1024 SIMDs x 2.4 GHz / 4" % bench_key}
        if sustained_mhz:   # ... and against the clock the part held under THIS load (ClockSampler: pp_dpm_sclk, sampled through the timed steps)
            peak_now = 1024 * sustained_mhz * 1e6 / 4
            r["valu_issue"].update({"sustained_mhz": sustained_mhz, "peak_at_sustained_clock": peak_now / 1e9,
                                    "frac_at_sustained_clock": per_s / peak_now})
    elif sustained_mhz:
        r["sustained_mhz"] = sustained_mhz
    return r


def run_stub(args, shard, rank, world):
    """Launcher / process-group / timing path without a GPU (tests/test_bench_launcher.py)."""
    dist, backend = shard.init_group(world, rank, 0, use_gpu=False)
    if args.global_log2_batch is not None:      # BASELINE config 5's form: contiguous slices of one global batch
        lo, hi = shard.shard_range(1 << args.global_log2_batch, rank, world)
        n, scaling = hi - lo, "strong"
    else:
        n, scaling, lo = 1 << args.log2_batch, "weak", 0
    step = lambda: time.sleep(args.stub_step_ms * 1e-3 * (1 + rank))
    mine, worst = shard.timed_region(step, args.steps, args.warmup, lambda: None, dist, backend)
    rows = shard.gather_over_ranks([rank, -1, n * args.steps / mine, mine / args.steps * 1e3, n, lo], dist, backend)
    if rank == 0:
        print(json.dumps({
            "metric": "stub steps", "value": sum(int(r[4]) for r in rows) * args.steps / worst, "unit": "ops/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": worst / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "none", "data": "stub",
            "config": {"workload": "stub", "backend": backend, "control_plane": backend},
            "per_gpu": [dict({"rank": int(r[0]), "device": int(r[1]), "value": r[2], "ms_per_step": r[3], "batch": int(r[4])},
                             **({"slice": [int(r[5]), int(r[5]) + int(r[4])]} if scaling == "strong" else {})) for r in rows]}),
            flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# the other BASELINE configs on the default line: (key, workload, table access of its calls)
# (key, workload, table access, base-table digits: 0 = the library's default, 20 bits / 2.2 GiB)
CONFIGS = (("fixed", "fixed", "index-independent", 0),          # config 3: a caller's comb table, staged in LDS
           ("base", "base", "index-independent", 0),            # ... the built-in base point, library default
           ("verify", "verify", "index-independent", 0),        # config 4 (public data: the mode changes nothing)
           ("verify_distinct_keys", "verify_distinct", "index-independent", 0),   # ... when no key repeats
           ("base_fast", "base", "fast", 0),
           ("verify_table24", "verify", "index-independent", 24))   # config 4 with the opt-in 24-bit table (28.5 GiB)


def run_rank(args):
    from libgoldilocks_amd import shard
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.stub_step_ms is not None:
        return run_stub(args, shard, rank, world)
    import numpy as np
    import torch
    import libgoldilocks_amd as ga
    visible = torch.cuda.device_count()          # counting devices does not initialise the GPU
    device = shard.device_for_rank(local_rank, visible)
    torch.cuda.set_device(device)
    dist, backend = shard.init_group(world, rank, visible)
    ga.lib()
    info = ga.device_info()

    if args.global_log2_batch is not None:
        lo, hi = shard.shard_range(1 << args.global_log2_batch, rank, world)
        n, scaling = hi - lo, "strong"
    else:
        n, scaling, lo = 1 << args.log2_batch, "weak", 0
    name = args.workload
    spec = WORKLOADS[name]
    cx = Ctx(ga, np, torch, n, rank)
    w = make_workload(name, cx, args.table_access)
    pci_of = torch.cuda.get_device_properties(torch.cuda.current_device())
    pci_txt = "%04x:%02x:%02x" % (getattr(pci_of, "pci_domain_id", 0), getattr(pci_of, "pci_bus_id", 0), getattr(pci_of, "pci_device_id", 0))
    clock = ClockSampler(pci_txt)
    mine, worst, kernel_ms = time_workload(torch, shard, w, args.steps, args.warmup, dist, backend, clock)
    if "kernels_after" in w:
        w["kernel"], w["traffic_kernels"], w["macs"] = w["kernels_after"]()
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    # (each rank's device by its PCI address: a first real 8-GPU run shows at a glance that eight ranks sit on eight devices)
    props = torch.cuda.get_device_properties(device)
    pci = [getattr(props, k, -1) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    rows = shard.gather_over_ranks([rank, device, n * args.steps / mine, avg_ms, n, lo] + pci, dist, backend)
    ok, check, extra = w["check"]() if rank == 0 else (True, "n/a", {})
    default_line = rank == 0 and world == 1 and name == "varbase" and args.table_access == "index-independent" \
        and args.global_log2_batch is None
    samples = {}

    line = None
    if rank == 0:
        total_ops = sum(int(r[4]) for r in rows) * args.steps
        batch_txt = ("2^%d" % (n.bit_length() - 1)) if n & (n - 1) == 0 else str(n)
        line = {
            "metric": "%s, batch=%s" % (spec["metric"], batch_txt), "value": total_ops / worst, "unit": spec["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": worst / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": spec["desc"], "batch_per_gpu": n, "table_access": args.table_access,
                       "table_access_is_library_default": args.table_access == "index-independent",
                       "sharding": ("contiguous slices of one global batch of 2^%d" % args.global_log2_batch
                                    if args.global_log2_batch is not None else "independent batch per GPU")
                                   + ", no data-path collective", "control_plane": backend or "single process",
                       # ranks whose RCCL bring-up succeeded (None: not attempted); RCCL carries the barrier only if all did
                       "rccl_ranks_seen": shard.RCCL_RANKS_SEEN[0],
                       "io_layout": "AoS reference structs resident in HBM", "device": info["arch"],
                       # digits of the base point's window table on this device (0: this workload never asked for it)
                       "base_table_bits": ga.get_base_table_bits(),
                       # what the library holds on the device after the timed steps (workspace + staging + tables)
                       "device_memory_bytes": ga.device_info()["workspace_bytes"],
                       "parity_spot_check": "ok" if ok else "FAILED", "check": check},
            # the code object this line was measured on and what compiled it (parity evidence is tied to both)
            "build": ga.build_info(),
            "per_gpu": [dict({"rank": int(r[0]), "device": int(r[1]), "value": r[2], "unit": spec["unit"],
                              "kernel_ms_avg": r[3], "batch": int(r[4]),
                              "pci": "%04x:%02x:%02x" % (int(r[6]), int(r[7]), int(r[8])) if r[6] >= 0 else None},
                             **({"slice": [int(r[5]), int(r[5]) + int(r[4])]} if scaling == "strong" else {})) for r in rows],
            "roofline": roofline(name, w["kernel"], n, avg_ms, args.table_access, w.get("traffic_kernels"), ga.get_base_table_bits(), w.get("macs"),
                                 clock.sustained_mhz()),
        }
        line["roofline"]["kernel_ms_every_step"] = [round(x, 3) for x in kernel_ms]   # (rank 0's; HIP events on the launch stream)
        line.update(extra)
        if "sample" in w and world == 1:
            samples["headline"] = w["sample"]()

    # the other BASELINE configs, a few steps each (default single-GPU run of the headline only)
    if default_line and not args.no_configs:
        configs = {}
        for key, cname, access, table_bits in CONFIGS:
            ga.set_base_table_bits(table_bits)
            cw = make_workload(cname, cx, access)
            # The clocks of a GPU that idled (building a workload's input leaves it idle for tens of milliseconds) take
            # 40 - 50 ms of load to come back up (profiles/r04/experiments.md G: a verification step behind 20 ms of
            # idling takes 9.3 ms instead of 7.6, and five steps to settle): three warm-up launches do for a 34-ms
            # kernel, not for a 2-ms one.  Warm up for 150 ms' worth of steps, time 8 (12 of the short ones).
            cw["step"]()
            torch.cuda.synchronize()
            t_probe = time.perf_counter()
            cw["step"]()
            torch.cuda.synchronize()
            t_probe = max(time.perf_counter() - t_probe, 1e-4)
            cwarm = max(3, min(100, int(0.15 / t_probe)))
            csteps = 8 if t_probe > 0.02 else 12
            cclock = ClockSampler(pci_txt)
            _, cworst, cms = time_workload(torch, shard, cw, csteps, cwarm, None, None, cclock)
            if "kernels_after" in cw:
                cw["kernel"], cw["traffic_kernels"], cw["macs"] = cw["kernels_after"]()
            cok, ctext, _ = cw["check"]()
            cavg = sum(cms) / len(cms)
            r = roofline(cname, cw["kernel"], n, cavg, access, cw.get("traffic_kernels"), ga.get_base_table_bits(), cw.get("macs"),
                         cclock.sustained_mhz())
            configs[key] = {"value": n * csteps / cworst, "unit": WORKLOADS[cname]["unit"], "workload": WORKLOADS[cname]["desc"],
                            "table_access": access, "steps": csteps, "warmup": cwarm, "kernel": cw["kernel"], "kernel_ms_avg": cavg,
                            "roofline": {k: r[k] for k in ("achieved", "frac", "traffic", "unit", "traffic_measured_on", "traffic_stale",
                                                               "traffic_of_base_table_gathers", "traffic_note") if k in r},
                            "mac_frac": r["mac"]["frac"] if "mac" in r else None, "macs_per_op": r["mac"]["macs_per_op"] if "mac" in r else None,
                            # the ceiling that binds, for every config: VALU issue at the nominal and at the sustained clock
                            "valu_issue": r.get("valu_issue"), "sustained_mhz": cclock.sustained_mhz(),
                            "base_table_bits": ga.get_base_table_bits(), "base_table_bits_is_library_default": table_bits == 0,
                            # what the library holds on the device now (workspace + staging + the base point's table)
                            "device_memory_bytes": ga.device_info()["workspace_bytes"], "check": ctext,
                            "parity_spot_check": "ok" if cok else "FAILED"}
            ok = ok and cok
            samples[key] = cw["sample"]()
            del cw
            if table_bits:                       # an opt-in table goes again: the next config sees the default's footprint
                ga.set_base_table_bits(0)
                ga.release_memory(ga.RELEASE_BASE_TABLE)
        line["configs"] = configs

    if default_line and not args.no_end_to_end:
        line["end_to_end"], e2e_ok = end_to_end(cx)
        ok = ok and e2e_ok

    if rank == 0 and world == 1 and name == "varbase" and not args.no_cpu_baseline:   # rank 0 at N = 1 only
        bases, scalars = cx.pairs()
        b_h, s_h = bases.cpu().numpy().view(np.uint64), scalars.cpu().numpy().view(np.uint64)
        line["cpu_baseline"], checks = cpu_baseline_leg(np, b_h, s_h, samples)
        for key, (same, text) in checks.items():
            ok = ok and same
            if key == "headline":
                line["config"]["check"] += "; " + text
            else:
                line["configs"][key]["check"] = text + "; " + line["configs"][key]["check"]
                if not same:
                    line["configs"][key]["parity_spot_check"] = "FAILED"
        line["config"]["parity_spot_check"] = "ok" if ok else "FAILED"

    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not ok:
        sys.exit(2)


def main(argv=None):
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: one fresh process per rank, started before this process has
        # touched a GPU (it never does: shard.py imports neither torch nor HIP)
        from libgoldilocks_amd import shard
        code = shard.launch_ranks([os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv), args.gpus)
        print(json.dumps({"launcher": {"ranks": args.gpus, "exit_code": code,
                                       "torch_imported_by_launcher": any(m == "torch" or m.startswith("torch.")
                                                                         for m in sys.modules)}}), file=sys.stderr)
        sys.exit(code)
    run_rank(args)


if __name__ == "__main__":
    main()
