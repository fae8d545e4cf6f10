"""Manual check (not collected by pytest): one GPU's share of BASELINE config 5 -- 2^21 verifications in one
launch (2^24 over 8 GPUs) -- and 2^23 of them, a tenth corrupted, exact accept / reject sets.  python tests/config5_check.py"""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, libgoldilocks_amd as ga, _gen

for lg in (21, 23):
    n = 1 << lg
    nk = 1024
    sk = np.frombuffer(_gen.stream(b"c5/sk", 57 * nk), np.uint8).reshape(nk, 57)
    sk_d = torch.from_numpy(np.ascontiguousarray(sk[np.arange(n) % nk])).cuda()
    msg = torch.from_numpy(np.frombuffer(_gen.stream(b"c5/msg", 32 * 4096), np.uint8).reshape(4096, 32)[np.arange(n) % 4096].copy()).cuda()
    idx = torch.arange(n, device="cuda")
    msg[:, 0] = (idx & 0xff).to(torch.uint8); msg[:, 1] = ((idx >> 8) & 0xff).to(torch.uint8); msg[:, 2] = ((idx >> 16) & 0xff).to(torch.uint8)
    pk = torch.empty((n, 57), dtype=torch.uint8, device="cuda")
    sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    ga.dev("ed448_derive_public_key", pk.data_ptr(), sk_d.data_ptr(), n, None)
    ga.dev("ed448_sign", sig.data_ptr(), sk_d.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    bad = (idx % 10) == 3
    sig[bad, 60] ^= 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert bool(((st == -1) == ~bad).all())
    print("2^%d verifications: %.1f ms, %.2f M/s, accept / reject sets exact" % (lg, dt * 1e3, n / dt / 1e6), flush=True)
