#!/usr/bin/env python3
"""bench.py -- Ed448 variable-base scalarmuls/s, batch 2^20 per GPU (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of goldilocks_448_point_scalarmul over a batch of 2^20 independent
(point, scalar) pairs that are already resident in HBM (AoS reference structs), i.e. one launch
of k_point_scalarmul through the C ABI (goldilocks_amd_point_scalarmul_dev).  Ranks own
independent batches (weak scaling, no collective on the data path); the only collectives are the
timing barrier and the MAX over ranks.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      HBM roofline of the dominant kernel: algorithmic bytes (568 B/op: 256 B point +
                56 B scalar in, 256 B point out) / average launch duration measured
                with HIP events on the launch stream, against 8 TB/s.  The path is integer-VALU
                bound, so this fraction is tiny by construction; "valu" carries the honest
                ceiling: achieved VALU wave-instructions/s vs the measured issue peak of the
                ladder's instruction mix, and 32x32->64 MAC/s vs the measured v_mad_u64_u32 peak.
  cpu_baseline  the REAL reference (arch_x86_64 path, oracle/_ref, built for generic x86-64) --
                or the oracle port if that .so did not travel -- timed on the host cores over a
                bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LOG2_BATCH = 20
BYTES_PER_OP = 568          # algorithmic: 256 (point in) + 56 (scalar in) + 256 (point out)
MACS_PER_OP = 680_584       # 2279 M x 192 + 1785 S x 136 + 16 mulw x 16 MACs per op
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_MAC_PEAK = 36.0e12     # measured: 560-576 G v_mad_u64_u32 wave-instr/s x 64 lanes (profiles/r01/ubench.txt)
VALU_INSTR_PER_OP = 1.092e6 # VALU wave-instructions per op: SQ_INSTS_VALU / wave-ops (profiles/r01/rocprofv3_pmc_summary.json)
VALU_ISSUE_PEAK = 600e9     # measured: wave-instr/s of a 1:1 MAC:simple mix at 2+ waves/SIMD (ubench mix_mac_add)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-batch", type=int, default=LOG2_BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="varbase", choices=["varbase", "fixed", "base", "verify", "sign", "x448", "direct"])
    ap.add_argument("--table-access", default="fast", choices=["fast", "index-independent"],
                    help="goldilocks_amd_set_table_access: how base-point tables are read for secret scalars "
                         "(affects the base and sign workloads)")
    return ap.parse_args()


def make_inputs(ga, np, torch, n, rank):
    """Synthetic batch from the SHAKE256 stream "bench_varbase_v1/<rank>/..." (tests/_gen.py):
    scalars uniform below 2^446; base points = k*B for stream scalars k (all distinct), computed
    on the device by the fixed-base kernel.  Rank 0's batch is the one whose outputs are pinned
    by tests/golden/f6_bench_digest.json (computed with the real reference)."""
    import _gen

    def stream_scalars(what):
        s = _gen.stream_scalars(n, b"bench_varbase_v1/%d/%s" % (rank, what))
        return torch.from_numpy(s.view(np.int64)).cuda()

    scalars, k = stream_scalars(b"scalar"), stream_scalars(b"base")
    bases = torch.empty((n, 32), dtype=torch.int64, device="cuda")
    ga.dev("precomputed_scalarmul", bases.data_ptr(), None, k.data_ptr(), n, None)
    torch.cuda.synchronize()
    return bases, scalars, k


def cpu_baseline(np, bases_h, scalars_h):
    """Time the reference's CPU path on the host cores over a bounded sample.  The only place in this
    file that touches oracle/ (test infrastructure): it returns the canonical encodings of the first
    256 reference results so that the caller can check the GPU's against them."""
    from _libs import oracle, REF_X86_SO
    O = oracle()
    cores = os.cpu_count() or 1
    threads = min(cores, 256)
    m = min(len(scalars_h), threads * 8192)
    b = np.ascontiguousarray(bases_h[:m])
    s = np.ascontiguousarray(scalars_h[:m])
    out = np.empty((m, 32), dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    kind, what = "port", "oracle/gold_oracle.c (ref64-shaped restatement)"
    fn = None
    if os.path.exists(REF_X86_SO):
        try:
            R = C.CDLL(REF_X86_SO)
            fn = C.cast(R.goldilocks_448_point_scalarmul, C.c_void_p)
            kind, what = "reference", "reference arch_x86_64 path (oracle/_ref, gcc -O2 generic x86-64)"
        except OSError:
            fn = None
    run = (lambda: O.orc_extern_scalarmul_batch(fn, p(out), p(b), p(s), m, threads)) if fn else \
          (lambda: O.orc_point_scalarmul_batch(p(out), p(b), p(s), m, threads))
    O.orc_point_scalarmul_batch(p(out), p(b), p(s), min(m, threads), threads)  # warm tables/threads
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    import _gen
    return {"value": m / dt, "unit": "scalarmuls/s", "cores": threads, "kind": kind,
            "sample": "%d of the 2^20 (point, scalar) pairs, %d threads, %.1f s; %s" % (m, threads, dt, what)}, \
        _gen.oracle_encode(out[:256])


def main():
    args = parse()
    import numpy as np
    import torch
    import libgoldilocks_amd as ga
    ga.set_table_access(ga.TABLES_INDEX_INDEPENDENT if args.table_access == "index-independent" else ga.TABLES_FAST)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("GOLDILOCKS_BENCH_FORCE_DIST"):   # the env knob lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("NCCL_DEBUG", "WARN")    # keep RCCL's banner off stdout: one JSON line only
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ga.lib()
    info = ga.device_info()

    n = 1 << args.log2_batch
    bases, scalars, _ = make_inputs(ga, np, torch, n, rank)
    out = torch.empty_like(bases)
    stream = torch.cuda.current_stream().cuda_stream

    if args.workload == "varbase":
        step = lambda: ga.dev("point_scalarmul", out.data_ptr(), bases.data_ptr(), scalars.data_ptr(), n, stream)
        bytes_per_op, kernel = BYTES_PER_OP, "k_point_scalarmul"
    elif args.workload == "fixed":     # BASELINE config 3: caller's precomputed_s -> the 5x5x18 comb staged in LDS
        comb_tab = torch.from_numpy(ga.precomputed_base().view(np.int64)).cuda()
        step = lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), comb_tab.data_ptr(), scalars.data_ptr(), n, stream)
        bytes_per_op, kernel = 312, "k_precomputed_scalarmul"
    elif args.workload == "base":      # the built-in base point: 16-bit window table (no doublings)
        step = lambda: ga.dev("precomputed_scalarmul", out.data_ptr(), None, scalars.data_ptr(), n, stream)
        bytes_per_op, kernel = 312, ("k_precomputed_scalarmul" if args.table_access == "index-independent" else "k_base_scalarmul")
    elif args.workload == "direct":    # wire format in and out: 56-byte encodings, decode + ladder + encode fused
        enc_in = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        ga.dev("point_encode", enc_in.data_ptr(), bases.data_ptr(), n, None)
        enc_out = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        st_direct = torch.empty(n, dtype=torch.int32, device="cuda")
        step = lambda: ga.dev("direct_scalarmul", enc_out.data_ptr(), st_direct.data_ptr(), enc_in.data_ptr(),
                              scalars.data_ptr(), 0, 0, n, stream)
        bytes_per_op, kernel = 56 + 56 + 56 + 4, "k_direct_scalarmul"
    elif args.workload in ("sign", "x448"):
        import _gen
        nb = 57 if args.workload == "sign" else 56
        sk = torch.from_numpy(np.frombuffer(_gen.stream(b"bench_%s_v1/%d/sk" % (args.workload.encode(), rank), nb * n),
                                            np.uint8).reshape(n, nb).copy()).cuda()
        if args.workload == "sign":
            pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
            ga.dev("ed448_derive_public_key", pk.data_ptr(), sk.data_ptr(), n, None)
            msg = torch.from_numpy(np.frombuffer(_gen.stream(b"bench_sign_v1/%d/msg" % rank, 32 * n), np.uint8)
                                   .reshape(n, 32).copy()).cuda()
            sig_out = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
            step = lambda: ga.dev("ed448_sign", sig_out.data_ptr(), sk.data_ptr(), pk.data_ptr(), msg.data_ptr(), None,
                                  32, 0, None, 0, n, stream)
            bytes_per_op, kernel = 57 + 57 + 32 + 114, ("k_ed448_sign_ct" if args.table_access == "index-independent" else "k_ed448_sign")
        else:
            pub = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
            ga.dev("x448", pub.data_ptr(), None, None, sk.data_ptr(), n, None)
            peer = pub.view(n // 2, 2, 56).flip(1).reshape(n, 56).contiguous()      # lane i meets lane i^1's public key
            shared = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
            st448 = torch.empty(n, dtype=torch.int32, device="cuda")
            step = lambda: ga.dev("x448", shared.data_ptr(), st448.data_ptr(), peer.data_ptr(), sk.data_ptr(), n, stream)
            bytes_per_op, kernel = 56 * 3 + 4, "k_x448"
    else:
        # 2^20 signatures over 32-byte messages from 1024 distinct keys, produced by
        # the library's own derive/sign kernels (bit-exact vs the reference: tests/test_gpu_parity.py)
        import _gen
        nk, nsig = 1024, 4096
        sk_k = np.frombuffer(_gen.stream(b"bench_verify_v1/%d/sk" % rank, 57 * nk), np.uint8).reshape(nk, 57)
        pk_k = ga.ed448_derive_public_key_batch(sk_k)
        key_of = np.arange(nsig) % nk
        msg_h = np.frombuffer(_gen.stream(b"bench_verify_v1/%d/msg" % rank, 32 * nsig), np.uint8).reshape(nsig, 32)
        sigs = ga.ed448_sign_batch(sk_k[key_of], pk_k[key_of], [m.tobytes() for m in msg_h])
        pks = pk_k[key_of]
        idx = np.random.default_rng(rank).integers(0, nsig, n)
        bad = np.random.default_rng(rank + 99).random(n) < 0.01           # 1 % corrupted signatures
        sig_h = sigs[idx]
        sig_h[bad, 5] ^= 0x20
        d_sig, d_pk = torch.from_numpy(sig_h).cuda(), torch.from_numpy(pks[idx]).cuda()
        d_msg = torch.from_numpy(msg_h[idx].copy()).cuda()
        status = torch.empty(n, dtype=torch.int32, device="cuda")
        step = lambda: ga.dev("ed448_verify", status.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(),
                              None, 32, 0, None, 0, n, stream)
        bytes_per_op, kernel = 175 + 32, "k_ed448_verify"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]   # HIP events on the launch stream
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Check of what was just timed (not in the timed region).  No oracle here: size-independent
    # properties on the device, the reference's digest fixture for the headline batch, and -- inside
    # the cpu_baseline leg only -- the reference's own outputs on its sample.
    ok = True
    check = "n/a"
    extra = {}
    if rank == 0 and args.workload == "sign":
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        ga.dev("ed448_verify", st.data_ptr(), sig_out.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
        ok = int((st == -1).sum()) == n
        check = "every signature verifies (ed448_verify kernel)"
    elif rank == 0 and args.workload == "x448":
        pairs = shared.view(n // 2, 2, 56)
        ok = bool((pairs[:, 0] == pairs[:, 1]).all()) and int((st448 == -1).sum()) == n
        check = "Diffie-Hellman symmetry: X448(a, pub_b) == X448(b, pub_a) for every neighbour pair"
    elif rank == 0 and args.workload == "direct":
        ref = torch.empty_like(bases)
        ga.dev("point_scalarmul", ref.data_ptr(), bases.data_ptr(), scalars.data_ptr(), n, None)
        ref_enc = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
        ga.dev("point_encode", ref_enc.data_ptr(), ref.data_ptr(), n, None)
        ok = bool((ref_enc == enc_out).all()) and int((st_direct == -1).sum()) == n
        check = "every output equals encode(point_scalarmul(decode(input))) computed by the separate kernels"
    elif rank == 0 and args.workload in ("fixed", "base"):
        m = 1 << 14
        base_pt = torch.from_numpy(np.repeat(ga.point_base().reshape(1, 32), m, axis=0).view(np.int64)).cuda()
        alt = torch.empty((m, 32), dtype=torch.int64, device="cuda")
        ga.dev("point_scalarmul", alt.data_ptr(), base_pt.data_ptr(), scalars.data_ptr(), m, None)
        st = torch.empty(m, dtype=torch.int32, device="cuda")
        ga.dev("point_pred", st.data_ptr(), alt.data_ptr(), out.data_ptr(), 0, m, None)
        ok = int((st == -1).sum()) == m
        check = "first 2^14 results equal the variable-base ladder applied to the base point"
    elif rank == 0 and args.workload == "varbase":
        b_h, s_h = bases.cpu().numpy().view(np.uint64), scalars.cpu().numpy().view(np.uint64)
        stv = torch.empty(n, dtype=torch.int32, device="cuda")
        ga.dev("point_pred", stv.data_ptr(), out.data_ptr(), None, 1, n, None)
        ok = int((stv == -1).sum()) == n
        check = "every output is a valid point"
        dig_path = os.path.join(ROOT, "tests", "golden", "f6_bench_digest.json")
        if os.path.exists(dig_path) and str(args.log2_batch) in json.load(open(dig_path))["digest_shake256_32"]:
            import hashlib
            ser = torch.empty((n, 56), dtype=torch.uint8, device="cuda")
            ga.dev("point_encode", ser.data_ptr(), out.data_ptr(), n, None)
            digest = hashlib.shake_256(ser.cpu().numpy().tobytes()).hexdigest(32)
            match = digest == json.load(open(dig_path))["digest_shake256_32"][str(args.log2_batch)]
            ok = ok and match
            check += "; SHAKE256 digest of all outputs equals the reference's (golden F6)"
            extra["full_batch_digest"] = {"shake256_32": digest, "matches_reference_fixture": match}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
            extra["cpu_baseline"], ref_enc = cpu_baseline(np, b_h, s_h)
            same = bool((ga.point_encode_batch(out[:256].cpu().numpy().view(np.uint64)) == ref_enc).all())
            ok = ok and same
            check += "; first 256 results bit-exact vs the CPU baseline's outputs"
    elif rank == 0:
        ok = abs(int((status == -1).sum()) - int((~bad).sum())) == 0
        check = "accepted == uncorrupted lanes (signatures made by the sign kernel)"

    if rank == 0:
        total_ops = n * args.steps * world
        value = total_ops / dt
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = bytes_per_op * n / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get(kernel)
        line = {
            "metric": {"varbase": "Ed448 variable-base scalarmuls/sec, batch=2^%d" % args.log2_batch,
                       "fixed": "Ed448 fixed-base scalarmuls/sec, batch=2^%d" % args.log2_batch,
                       "base": "Ed448 base-point scalarmuls/sec, batch=2^%d" % args.log2_batch,
                       "verify": "Ed448 verifies/sec, batch=2^%d" % args.log2_batch, "sign": "Ed448 signatures/sec, batch=2^%d" % args.log2_batch,
                       "x448": "X448 shared secrets/sec, batch=2^%d" % args.log2_batch,
                       "direct": "Ed448 wire-format scalarmuls/sec, batch=2^%d" % args.log2_batch}[args.workload],
            "value": value, "unit": {"varbase": "scalarmuls/s", "fixed": "scalarmuls/s", "base": "scalarmuls/s", "verify": "verifies/s",
                                     "sign": "signatures/s", "x448": "shared secrets/s", "direct": "scalarmuls/s"}[args.workload],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": {"varbase": "goldilocks_448_point_scalarmul, variable base, random scalars",
                                    "fixed": "goldilocks_448_precomputed_scalarmul, 5x5x18 comb table staged in LDS",
                                    "base": "goldilocks_448_precomputed_scalarmul(precomputed_base), 16-bit window table",
                                    "verify": "goldilocks_ed448_verify, 32-byte messages, 1% corrupted",
                                    "sign": "goldilocks_ed448_sign, 32-byte messages, no context",
                                    "x448": "goldilocks_x448, random peer public keys",
                                    "direct": "goldilocks_448_direct_scalarmul, 56-byte encodings in and out"}[args.workload],
                       "batch_per_gpu": n, "table_access": args.table_access,
                       "sharding": "independent batch per GPU, no data-path collective",
                       "io_layout": "AoS reference structs resident in HBM", "device": info["arch"],
                       "parity_spot_check": "ok" if ok else "FAILED", "check": check},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel,
                         "kernel_ms_avg": avg_ms, "bytes_per_op": bytes_per_op,
                         "note": "integer-VALU bound by construction; see valu"},
        }
        if args.workload == "varbase":
            macs = MACS_PER_OP * n / (avg_ms * 1e-3)
            issue = VALU_INSTR_PER_OP * n / 64 / (avg_ms * 1e-3)
            line["valu"] = {"bound": "VALU issue (every VALU op costs ~4 SIMD-cycles once interleaved with MACs)",
                            "achieved": issue / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instr/s",
                            "frac": issue / VALU_ISSUE_PEAK,
                            "mac": {"achieved": macs / 1e12, "peak": VALU_MAC_PEAK / 1e12, "unit": "T MAC/s",
                                    "frac": macs / VALU_MAC_PEAK}}
        line.update(extra)
        print(json.dumps(line), flush=True)
        if not ok:
            sys.exit(2)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
