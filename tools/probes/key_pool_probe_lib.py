"""make() / timeit() shared by the verification timing probes."""
import numpy as np, torch, libgoldilocks_amd as ga, _gen
def make(n, nk):
    sk = np.frombuffer(_gen.stream(b"pp/sk", 57 * nk), np.uint8).reshape(nk, 57)
    pk_k = ga.ed448_derive_public_key_batch(sk)
    key_of = np.random.default_rng(5).integers(0, nk, n)
    msg = np.frombuffer(_gen.stream(b"pp/msg", 32 * n), np.uint8).reshape(n, 32).copy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_sk, d_pk, d_msg = d(sk[key_of]), d(pk_k[key_of]), d(msg)
    sig = torch.empty((n, 114), dtype=torch.uint8, device="cuda")
    ga.dev("ed448_sign", sig.data_ptr(), d_sk.data_ptr(), d_pk.data_ptr(), d_msg.data_ptr(), None, 32, 0, None, 0, n, None)
    return sig, d_pk, d_msg
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
