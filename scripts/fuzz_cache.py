"""randomised parity sweep of the cache-live prologue (offsets -> bag rows, cache lookup, stable partition) and of
the cache gather / SGD scatter, GPU vs CPU oracle.  usage: python scripts/fuzz_cache.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fbtt-embedding_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle_lib as O, tt_embeddings as E
from util import assert_close

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    H = int(rs.choice([64, 4096, 1 << 16, 1 << 20]))
    E_ = int(rs.choice([50, 5000, 11_000_000]))
    a = float(rs.choice([1.05, 1.3, 2.0]))
    keys, freq = np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)
    for _ in range(int(rs.randint(1, 4))):
        O.update_cache_state((rs.zipf(a, size=int(rs.randint(1, 5000))) % E_).astype(np.int64), keys, freq)
    cs = int(rs.randint(1, 2000))
    state = np.where((keys != -1) & (rs.rand(H) < 0.6), rs.randint(0, cs, size=H), -1).astype(np.int32)
    nnz = int(rs.choice([1, 63, 64, 255, 256, 257, 1000, 4096, 20000, 70000]))
    B = int(rs.choice([1, 5, 64, 512, 3000]))
    B = max(B, nnz // 300)  # (a row hit m times in a bag takes m*g here and g m times in the oracle: keep m modest)
    lens = rs.multinomial(nnz, np.ones(B) / B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = (rs.zipf(a, size=nnz) % E_).astype(np.int64)
    exp = O.preprocess_indices(idx, off, 1, False, keys, state)
    got = E.preprocess_indices_sync(t(idx), t(off), 1, False, t(keys), t(state))
    what = f"case {n}: H={H} nnz={nnz} B={B}"
    assert got[3] == exp[3], what + " num_tt"
    ntt = exp[3]
    for k, name in ((0, "colidx"), (1, "rowidx"), (2, "tableidx")):
        assert np.array_equal(got[k].cpu().numpy(), exp[k]), what + " " + name
    if ntt < nnz:
        assert np.array_equal(got[4].cpu().numpy()[ntt:], exp[4][ntt:]), what + " cache locations"
        D = int(rs.choice([4, 64, 60, 128, 7]))
        loc, rowidx = exp[4][ntt:].astype(np.int32), exp[1][ntt:]
        w = rs.randn(cs, D).astype(np.float32)
        out0 = rs.randn(1, B, D).astype(np.float32)
        ref = out0.copy()
        O.cache_forward(B, loc, rowidx, w, ref[0])
        dout = t(out0)
        E.cache_forward(B, nnz - ntt, t(loc), t(rowidx), t(w), dout)
        assert_close(dout.cpu().numpy(), ref, what + f" cache_forward D={D}", rtol=1e-4, atol_scale=2e-5)
        grad = (rs.rand(B, D) * 0.1).astype(np.float32)
        # reference in float64 (a row can take tens of thousands of adds here: the fp32 oracle's own sequential
        # rounding is then larger than the GPU's, whose partial sums are shorter)
        w_ref = w.astype(np.float64)
        np.subtract.at(w_ref, loc, 0.1 * grad[rowidx].astype(np.float64))
        dw = t(w)
        E.cache_backward_sgd(nnz - ntt, t(grad), t(loc), t(rowidx), 0.1, dw)
        assert_close(dw.cpu().numpy(), w_ref, what + f" cache_backward_sgd D={D}", rtol=3e-4, atol_scale=2e-5)
    n += 1
print(f"{n} cases ok in {time.time() - t0:.0f} s")
