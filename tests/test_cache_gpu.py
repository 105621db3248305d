"""GPU parity of the software-cache path (hash table, lookup + stable partition,
cache gather forward/backward, populate) against the CPU oracle.  The reference
has NO test for this path; bit-exactness is defined against the oracle's
sequential order (SURVEY.md section 7 'Hash parity')."""
import numpy as np
import pytest
import torch

import gen_inputs as G
import oracle_lib as O
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def zipf_indices(rs, n, E, a=1.2):
    return (rs.zipf(a, size=n).astype(np.int64)) % E


def test_update_cache_state_bit_exact_when_no_probe_conflict():
    """every key lands on its home slot (checked with the oracle) -> the table
    contents cannot depend on insertion order -> must equal the oracle bit for bit"""
    import tt_embeddings as E

    rs = np.random.RandomState(0)
    H = 1 << 20
    for trial in range(3):
        idx = zipf_indices(rs, 20000, 11_000_000) if trial else rs.randint(0, 11_000_000, size=20000).astype(np.int64)
        keys = np.full(H, -1, dtype=np.int64)
        freq = np.zeros(H, dtype=np.int64)
        homes = {}
        ok = []
        for k in idx:  # keep only keys whose home slot is theirs alone
            h = O.hash64(int(k), H)
            if homes.setdefault(h, int(k)) == int(k):
                ok.append(int(k))
        idx = np.array(ok, dtype=np.int64)
        O.update_cache_state(idx, keys, freq)
        dk, df = t(np.full(H, -1, dtype=np.int64)), t(np.zeros(H, dtype=np.int64))
        E.update_cache_state(t(idx), dk, df)
        assert np.array_equal(dk.cpu().numpy(), keys) and np.array_equal(df.cpu().numpy(), freq)
        # a second batch on top of the first (existing keys accumulate)
        O.update_cache_state(idx[::2], keys, freq)
        E.update_cache_state(t(idx[::2].copy()), dk, df)
        assert np.array_equal(dk.cpu().numpy(), keys) and np.array_equal(df.cpu().numpy(), freq)


def test_update_cache_state_under_collisions():
    """tiny table, heavy probing and overflow: slot assignment among racing new
    keys is order dependent (as in the reference), so check the order-free
    invariants: every stored key sits within 3 probes of its home slot, holds
    its full count, and counts stored + dropped == lookups; keys the oracle
    stores and the GPU stores agree whenever nothing was dropped."""
    import tt_embeddings as E

    rs = np.random.RandomState(1)
    for H, nkeys in ((64, 40), (257, 200), (1024, 1500)):
        idx = rs.randint(0, nkeys, size=5000).astype(np.int64) * 7919
        dk, df = t(np.full(H, -1, dtype=np.int64)), t(np.zeros(H, dtype=np.int64))
        E.update_cache_state(t(idx), dk, df)
        k, f = dk.cpu().numpy(), df.cpu().numpy()
        cnt = {}
        for v in idx:
            cnt[int(v)] = cnt.get(int(v), 0) + 1
        stored = 0
        seen = set()
        for slot in range(H):
            if k[slot] == -1:
                assert f[slot] == 0
                continue
            key = int(k[slot])
            assert key not in seen, "a key may not occupy two slots after one batch"
            seen.add(key)
            assert (slot - O.hash64(key, H)) % H < 3
            assert f[slot] == cnt[key]
            stored += int(f[slot])
        dropped = sum(c for key, c in cnt.items() if key not in seen)
        assert stored + dropped == idx.size
        ok_, of_ = np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)
        O.update_cache_state(idx, ok_, of_)
        assert abs(len(seen) - int((ok_ >= 0).sum())) <= max(2, nkeys // 20)


def _warm_table(rs, E_, H, batches=6, n=4000):
    keys = np.full(H, -1, dtype=np.int64)
    freq = np.zeros(H, dtype=np.int64)
    for _ in range(batches):
        O.update_cache_state(zipf_indices(rs, n, E_, 1.3), keys, freq)
    return keys, freq


@pytest.mark.parametrize("cache_size,H", [(64, 4096), (1000, 1 << 16), (300, 512)])
def test_cache_populate_parity(cache_size, H):
    import tt_embeddings as E

    p, q, r = [20, 22, 25], [4, 4, 4], [1, 16, 16, 1]
    E_, D = 11000, 64
    rs = np.random.RandomState(cache_size)
    cores = G.make_cores(3, 1, p, q, r)
    keys, freq = _warm_table(rs, E_, H)
    state = np.full(H, -1, dtype=np.int32)
    w = np.zeros((cache_size, D), dtype=np.float32)
    dk, df, ds, dw = t(keys), t(freq), t(state), t(w)
    O.cache_populate(O.make_geom(1, p, q, r), cores, keys, freq, state, w)
    E.cache_populate(E_, p, q, r, [t(c) for c in cores], torch.zeros(3, dtype=torch.int64, device=DEV), dk, df, ds, dw)
    assert np.array_equal(dk.cpu().numpy(), keys), "hashtbl after eviction"
    assert np.array_equal(df.cpu().numpy(), freq), "cache_freq after eviction"
    assert np.array_equal(ds.cpu().numpy(), state), "cache_state (slot -> cache row)"
    assert_close(dw.cpu().numpy(), w, "decompressed cache rows")
    # a SECOND populate on top of a stale cache_state, after the table has kept counting
    # (surviving keys keep their slots and counts, new hot keys enter, evicted slots are reused; cache rows
    # are re-decompressed from the cores, which were moved in between).  Keys chosen so that no two new keys
    # race for a slot -> the table must equal the oracle's bit for bit.
    more = zipf_indices(rs, 3000, E_, 1.1)
    homes, keep = {}, []
    for k in more:
        h = O.hash64(int(k), H)
        if homes.setdefault(h, int(k)) == int(k):
            keep.append(k)
    more = np.array(keep, dtype=np.int64)
    sim_k, sim_f = keys.copy(), freq.copy()
    O.update_cache_state(more, sim_k, sim_f)
    E.update_cache_state(t(more), dk, df)
    if np.array_equal(dk.cpu().numpy(), sim_k):  # (a probe chain through an evicted slot can still be order dependent)
        keys, freq = sim_k, sim_f
        cores2 = [cc * 1.5 for cc in cores]
        O.cache_populate(O.make_geom(1, p, q, r), cores2, keys, freq, state, w)
        E.cache_populate(E_, p, q, r, [t(cc) for cc in cores2], torch.zeros(3, dtype=torch.int64, device=DEV), dk, df, ds, dw)
        assert np.array_equal(dk.cpu().numpy(), keys) and np.array_equal(df.cpu().numpy(), freq), "second populate: table"
        assert np.array_equal(ds.cpu().numpy(), state), "second populate: cache_state"
        assert_close(dw.cpu().numpy(), w, "second populate: cache rows")
    else:
        assert int(df.sum()) >= int(freq.sum())


def test_preprocess_partition_bit_exact():
    import tt_embeddings as E

    p, q, r = [20, 22, 25], [4, 4, 4], [1, 16, 16, 1]
    E_, H, cs = 11000, 1 << 14, 500
    rs = np.random.RandomState(5)
    cores = G.make_cores(3, 1, p, q, r)
    keys, freq = _warm_table(rs, E_, H)
    state = np.full(H, -1, dtype=np.int32)
    w = np.zeros((cs, 64), dtype=np.float32)
    O.cache_populate(O.make_geom(1, p, q, r), cores, keys, freq, state, w)
    # (256, 257: the 256-position units of the partition; 300000: more than 1024 units -> the scan launch)
    for n, B in ((1, 1), (63, 7), (64, 8), (256, 9), (257, 9), (4097, 100), (20000, 512), (300000, 2000)):
        idx = zipf_indices(rs, n, E_, 1.3)
        lens = rs.multinomial(n, np.ones(B) / B)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        exp = O.preprocess_indices(idx, off, 1, False, keys, state)
        got = E.preprocess_indices_sync(t(idx), t(off), 1, False, t(keys), t(state))
        assert got[3] == exp[3], "num_tt"
        ntt = exp[3]
        assert np.array_equal(got[0].cpu().numpy(), exp[0]), "partitioned colidx (front in order, rear reversed)"
        assert np.array_equal(got[1].cpu().numpy(), exp[1]), "partitioned rowidx"
        assert np.array_equal(got[2].cpu().numpy(), exp[2]), "tableidx (not partitioned)"
        assert np.array_equal(got[4].cpu().numpy()[ntt:], exp[4][ntt:]), "cache locations of the cached tail"
        # warm-up / multi-table: no lookup
        g2 = E.preprocess_indices_sync(t(idx), t(off), 1, True, t(keys), t(state))
        assert g2[3] == n and g2[4] is None and np.array_equal(g2[0].cpu().numpy(), idx)
    e = torch.empty(0, dtype=torch.int64, device=DEV)
    g0 = E.preprocess_indices_sync(e, t(np.zeros(5, dtype=np.int64)), 1, False, t(keys), t(state))
    assert g0[3] == 0 and g0[4] is None


def test_cache_gather_forward_backward():
    import tt_embeddings as E

    rs = np.random.RandomState(9)
    cs, D, B = 300, 64, 40
    for D in (64, 60, 3):
        w = rs.randn(cs, D).astype(np.float32)
        n = 700
        rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
        loc = rs.randint(0, cs, size=n).astype(np.int32)
        out0 = rs.randn(1, B, D).astype(np.float32)
        exp = out0.copy()
        O.cache_forward(B, loc, rowidx, w, exp[0])
        dout = t(out0)
        E.cache_forward(B, n, t(loc), t(rowidx), t(w), dout)
        assert_close(dout.cpu().numpy(), exp, f"cache_forward D={D}")
        grad = (rs.rand(B, D) * 0.1).astype(np.float32)
        w_sgd = w.copy()
        O.cache_backward_sgd(grad, loc, rowidx, 0.1, w_sgd)
        dw = t(w)
        E.cache_backward_sgd(n, t(grad), t(loc), t(rowidx), 0.1, dw)
        assert_close(dw.cpu().numpy(), w_sgd, f"cache_backward_sgd D={D}")
        gd = E.cache_backward_dense(n, t(grad), t(loc), t(rowidx), 0.1, t(w))
        assert_close(gd.cpu().numpy(), O.cache_backward_dense(grad, loc, rowidx, cs, D), f"cache_backward_dense D={D}")
        # row-wise adagrad: deterministic when every cache row is hit by at most one lookup
        loc_u = rs.permutation(cs)[:200].astype(np.int32)
        row_u = np.sort(rs.randint(0, B, size=200)).astype(np.int64)
        st, w_a = (rs.rand(cs) * 0.01).astype(np.float32), w.copy()
        dst, dwa = t(st), t(w)
        O.cache_backward_rowwise_adagrad_approx(grad, loc_u, row_u, 0.1, 1e-4, st, w_a)
        E.cache_backward_rowwise_adagrad_approx(200, t(grad), t(loc_u), t(row_u), 0.1, 1e-4, dst, dwa)
        assert_close(dst.cpu().numpy(), st, f"rowwise adagrad state D={D}")
        assert_close(dwa.cpu().numpy(), w_a, f"rowwise adagrad weights D={D}")
        # rows hit from several bags: the state total is order independent
        st2, dst2 = np.zeros(cs, dtype=np.float32), t(np.zeros(cs, dtype=np.float32))
        O.cache_backward_rowwise_adagrad_approx(grad, loc, rowidx, 0.1, 1e-4, st2, w.copy())
        E.cache_backward_rowwise_adagrad_approx(n, t(grad), t(loc), t(rowidx), 0.1, 1e-4, dst2, t(w))
        assert_close(dst2.cpu().numpy(), st2, f"rowwise adagrad state total D={D}")


@pytest.mark.parametrize("n,D", [(5000, 64), (20000, 128), (300000, 64), (5000, 6)])
def test_cache_backward_hot_rows(n, D):
    """a Zipf stream over the cache rows (row 0 takes a sixth of the lookups): the rows [0, K) summed by their
    own work-groups (K = 64, 55 at 300k lookups, 0 for D % 4 != 0) and the per-lookup path for the rest give
    the oracle's SGD / dense result; also behind a device-side split point (skip_dev)."""
    import ctypes as C

    import tt_embeddings as E

    rs = np.random.RandomState(n + D)
    cs, B = 1000, 512
    loc = ((rs.zipf(1.2, size=n) - 1) % cs).astype(np.int32)
    rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
    w = rs.randn(cs, D).astype(np.float32)
    grad = (rs.rand(B, D) * 0.1).astype(np.float32)
    w_sgd = w.copy()
    O.cache_backward_sgd(grad, loc, rowidx, 0.1, w_sgd)
    dw = t(w)
    E.cache_backward_sgd(n, t(grad), t(loc), t(rowidx), 0.1, dw)
    # (the hottest row's update is a sum of n/6 terms, in an order of its own)
    assert_close(dw.cpu().numpy(), w_sgd, "cache_backward_sgd, hot rows", rtol=1e-4, atol_scale=2e-5)
    gd = E.cache_backward_dense(n, t(grad), t(loc), t(rowidx), 0.1, t(w))
    assert_close(gd.cpu().numpy(), O.cache_backward_dense(grad, loc, rowidx, cs, D), "cache_backward_dense, hot rows",
                 rtol=1e-4, atol_scale=2e-5)
    # the same entries behind a split point that lives in device memory
    k = 777
    L = E.lib()
    L.ttx_cache_backward_sgd_n.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_void_p, C.c_void_p]
    loc2 = t(np.concatenate([np.full(k, -1, dtype=np.int32), loc]))
    row2 = t(np.concatenate([np.zeros(k, dtype=np.int64), rowidx]))
    skip = torch.tensor([k], dtype=torch.int32, device=DEV)
    dw2, g = t(w), t(grad)
    assert L.ttx_cache_backward_sgd_n(n + k, skip.data_ptr(), D, g.data_ptr(), loc2.data_ptr(), row2.data_ptr(), 0.1,
                                      dw2.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0, L.ttx_last_error()
    assert_close(dw2.cpu().numpy(), w_sgd, "cache_backward_sgd_n, hot rows", rtol=1e-4, atol_scale=2e-5)


@pytest.mark.parametrize("n,D,cs", [(900, 64, 40), (1024, 128, 64), (700, 60, 64), (20000, 64, 1000)])
def test_cache_backward_rowwise_adagrad_hot_rows(n, D, cs):
    """row-wise Adagrad on a Zipf stream over the cache rows: rows [0, K) are served by their own work-groups, which
    take a segment's lookups of the row in INDEX order -- with the batch inside one segment (n <= 1024) and every row
    hot (cs <= 64, D % 4 == 0) that is the oracle's sequential order: state and weights match it; D % 4 != 0 keeps the
    per-lookup path, larger batches / colder rows are held to what does not depend on the order (the state totals)"""
    import tt_embeddings as E

    rs = np.random.RandomState(n + D)
    B = 256
    loc = ((rs.zipf(1.2, size=n) - 1) % cs).astype(np.int32)
    rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
    w = rs.randn(cs, D).astype(np.float32)
    grad = (rs.rand(B, D) * 0.1).astype(np.float32)
    st, w_o = (rs.rand(cs) * 0.01).astype(np.float32), w.copy()
    dst, dw = t(st), t(w)
    O.cache_backward_rowwise_adagrad_approx(grad, loc, rowidx, 0.1, 1e-4, st, w_o)
    E.cache_backward_rowwise_adagrad_approx(n, t(grad), t(loc), t(rowidx), 0.1, 1e-4, dst, dw)
    assert_close(dst.cpu().numpy(), st, "rowwise adagrad state totals", rtol=2e-5, atol_scale=4e-6)
    if n <= 1024 and cs <= 64 and D % 4 == 0:
        assert_close(dw.cpu().numpy(), w_o, "rowwise adagrad weights, sequential order", rtol=2e-5, atol_scale=4e-6)
    else:
        assert bool(torch.isfinite(dw).all())
        # every row moved against its gradients by at least the smallest and at most the largest possible step
        moved = np.abs(dw.cpu().numpy() - w).sum(axis=1)
        hit = np.bincount(loc, minlength=cs) > 0
        assert (moved[hit] > 0).all() and (moved[~hit] == 0).all()


@pytest.mark.parametrize("n,D,cs,B", [(900, 64, 40, 256), (5000, 64, 1000, 512), (20000, 128, 1000, 512), (300000, 64, 4096, 4096),
                                      (5000, 6, 300, 128), (70000, 60, 20000, 2048)])
def test_sorted_cache_update_is_deterministic_and_matches_the_oracle(n, D, cs, B):
    """Round 6: the cache rows' update WITHOUT atomics (ttx_cache_backward_sorted; the shim's `deterministic=True`): the cached
    lookups grouped by cache row with a stable sort, a row's bag gradients added in INDEX order, one writer per row.  On a Zipf
    stream over the cache rows (row 0 takes a sixth of the batch) for SGD, the dense gradient and row-wise Adagrad:
      * two runs are BIT-identical (the atomic kernels are not: that is what this replaces);
      * SGD / dense equal the float64 sum at the DEFAULT tolerance, hot rows included (the atomic path needs 10x for them);
      * row-wise Adagrad equals the sequential oracle -- state AND weights, whatever the batch size: index order within a row IS
        the oracle's order -- and the independent float64 restatement of the reference's kernel (util.rowwise_adagrad_segments_f64);
      * a device-side split point (skip_dev) in front of the cached lookups, and lookups with no cache row (-1) are left alone."""
    import tt_embeddings as E
    from util import rowwise_adagrad_segments_f64

    rs = np.random.RandomState(n + D)
    loc = ((rs.zipf(1.2, size=n) - 1) % cs).astype(np.int32)
    rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
    w = rs.randn(cs, D).astype(np.float32)
    grad = ((rs.rand(B, D) - 0.3) * 0.1).astype(np.float32)
    g64 = grad.astype(np.float64)
    # float64 sums per cache row
    delta = np.zeros((cs, D))
    np.add.at(delta, loc, g64[rowidx])
    dg, dl, dr = t(grad), t(loc), t(rowidx)
    # SGD
    runs = []
    for _ in range(2):
        dw = t(w)
        E.cache_backward_sgd(n, dg, dl, dr, 0.1, dw, deterministic=True)
        runs.append(dw.cpu().numpy())
    assert np.array_equal(runs[0], runs[1]), "sorted SGD update differs from run to run"
    # (a row's update is the fp32 sum of up to n / 6 gradient rows in a fixed tree of 32-position slices: against the float64 sum
    #  its error grows like sqrt(terms) ulps of the running sum; atol is tied to the largest |delta|)
    assert_close(runs[0], (w.astype(np.float64) - 0.1 * delta), "sorted cache_backward_sgd vs float64",
                 **(dict(rtol=2e-5, atol_scale=4e-6) if n >= 100000 else {}))
    # dense
    gd = [E.cache_backward_dense(n, dg, dl, dr, 0.1, t(w), deterministic=True).cpu().numpy() for _ in range(2)]
    assert np.array_equal(gd[0], gd[1])
    assert_close(gd[0], delta, "sorted cache_backward_dense vs float64", **(dict(rtol=2e-5, atol_scale=4e-6) if n >= 100000 else {}))
    assert not gd[0][np.bincount(loc, minlength=cs) == 0].any(), "rows nobody hit must stay zero"
    # row-wise Adagrad: the sequential oracle and the float64 restatement of the reference's kernel
    st0 = (rs.rand(cs) * 0.01).astype(np.float32)
    st_o, w_o = st0.copy(), w.copy()
    O.cache_backward_rowwise_adagrad_approx(grad, loc, rowidx, 0.1, 1e-4, st_o, w_o)
    ada = []
    for _ in range(2):
        dst, dwa = t(st0), t(w)
        E.cache_backward_rowwise_adagrad_approx(n, dg, dl, dr, 0.1, 1e-4, dst, dwa, deterministic=True)
        ada.append((dst.cpu().numpy(), dwa.cpu().numpy()))
    assert np.array_equal(ada[0][0], ada[1][0]) and np.array_equal(ada[0][1], ada[1][1]), "sorted Adagrad differs from run to run"
    tol = dict(rtol=2e-5, atol_scale=4e-6)  # (state: a running fp32 sum of up to n / 6 terms, associated differently from the oracle's)
    assert_close(ada[0][0], st_o, "sorted rowwise adagrad state vs oracle", **tol)
    # (weights: the hottest row takes n / 6 sequential steps, each with its own rounding of the running state -- at 300k lookups the
    #  50k updates of row 0 drift 4e-5 apart between the two associations of the state's prefix sum)
    assert_close(ada[0][1], w_o, "sorted rowwise adagrad weights vs oracle", **(dict(rtol=1e-4, atol_scale=4e-6) if n >= 100000 else tol))
    if n <= 20000:
        st64, w64 = rowwise_adagrad_segments_f64(grad, loc, rowidx, 0.1, 1e-4, st0, w)
        assert_close(st_o, st64, "oracle rowwise adagrad state vs the float64 restatement of cu:1735-1795", **tol)
        assert_close(w_o, w64, "oracle rowwise adagrad weights vs the float64 restatement", **tol)
        assert_close(ada[0][1], w64, "sorted rowwise adagrad weights vs the float64 restatement", **tol)
    # behind a device-side split point, with uncached entries (-1) in front of it
    import ctypes as C
    k = 777
    loc2 = t(np.concatenate([np.full(k, -1, dtype=np.int32), loc]))
    row2 = t(np.concatenate([np.zeros(k, dtype=np.int64), rowidx]))
    skip = torch.tensor([k], dtype=torch.int32, device=DEV)
    dw2 = t(w)
    E._cache_backward_sorted(E.OPTIM_SGD, n + k, dg, loc2, row2, 0.1, 0.0, None, dw2, skip_dev=skip)
    assert np.array_equal(dw2.cpu().numpy(), runs[0]), "the split point must not change a bit"


@pytest.mark.parametrize("n,D,cs,what", [
    (1, 64, 8, "one"), (700, 64, 1, "one_row"), (900, 64, 50, "none_cached"), (900, 64, 50, "skip_all"),
    (40000, 64, 50, "none_cached"), (3000, 256, 30, "plain"), (3000, 260, 30, "plain"), (5000, 8, 3, "plain"),
    (33000, 64, 5, "plain"), (2500, 128, 4_000_000, "plain"), (2500, 64, 9_000_000, "plain")])
def test_sorted_cache_update_edge_cases(n, D, cs, what):
    """the atomic-free update at the edges of its two routes (one launch up to 32,768 lookups with D % 4 == 0, D <= 256 and at most
    4096 x 4096 cache rows; the sort chain otherwise): a single lookup, one cache row taking the whole batch (several rounds of the
    owner's list), nothing cached (all -1 / the split point behind the batch: nothing may change), the widest row the one-launch
    kernel takes and the first it does not, a row of two float4, a batch just beyond the one-launch limit, caches of millions of
    rows (more work-groups / the chain).  SGD, dense and row-wise Adagrad against float64 / the oracle, twice, bit-identical."""
    import tt_embeddings as E

    rs = np.random.RandomState(n + D + cs % 1000)
    B = 64
    loc = ((rs.zipf(1.3, size=n) - 1) % cs).astype(np.int32) if cs > 1 else np.zeros(n, dtype=np.int32)
    if cs > 1_000_000:
        loc = rs.randint(0, cs, size=n).astype(np.int32)
        loc[: n // 3] = loc[0]  # (one hot row among millions)
    rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
    if what == "none_cached":
        loc[:] = -1
    grad = ((rs.rand(B, D) - 0.4) * 0.1).astype(np.float32)
    rows = np.unique(loc[loc >= 0])
    w0 = {int(r_): rs.randn(D).astype(np.float32) for r_ in rows}
    dw = torch.zeros(cs, D, device=DEV)
    if rows.size:
        dw[torch.from_numpy(rows.astype(np.int64)).to(DEV)] = t(np.stack([w0[int(r_)] for r_ in rows]))
    dst = torch.full((cs,), 0.01, device=DEV)
    skip = torch.tensor([n if what == "skip_all" else 0], dtype=torch.int32, device=DEV)
    live = (loc >= 0) & (what != "skip_all")
    delta = {}
    for i in np.nonzero(live)[0]:
        delta.setdefault(int(loc[i]), np.zeros(D))
        delta[int(loc[i])] += grad[rowidx[i]].astype(np.float64)
    for optim in (E.OPTIM_SGD, E.OPTIM_DENSE, E.OPTIM_ADAGRAD):
        outs = []
        for _ in range(2):
            w = dw.clone()
            st = dst.clone()
            target = torch.full((cs, D), 7.0, device=DEV) if optim == E.OPTIM_DENSE else w
            E._cache_backward_sorted(optim, n, t(grad), t(loc), t(rowidx), 0.1, 1e-4, st if optim == E.OPTIM_ADAGRAD else None, target,
                                     skip_dev=skip)
            outs.append((target, st))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "two runs differ"
        got = outs[0][0]
        touched = torch.zeros(cs, dtype=torch.bool, device=DEV)
        if delta:
            touched[torch.tensor(sorted(delta), device=DEV)] = True
        if optim == E.OPTIM_DENSE:
            assert not got[~touched].any(), "dense gradient of rows nobody hit must be zero"
        else:
            assert torch.equal(got[~touched], dw[~touched]), "rows nobody hit must not change"
        tol = dict(rtol=5e-5, atol_scale=1e-5) if n >= 30000 or cs == 1 else {}
        for r_, dv in list(delta.items())[:200]:
            if optim == E.OPTIM_SGD:
                assert_close(got[r_].cpu().numpy(), w0[r_].astype(np.float64) - 0.1 * dv, f"sgd row {r_}", **tol)
            elif optim == E.OPTIM_DENSE:
                assert_close(got[r_].cpu().numpy(), dv, f"dense row {r_}", **tol)
        if optim == E.OPTIM_ADAGRAD and delta:
            st_o = np.full(cs if cs <= 100000 else 1, 0.01, dtype=np.float32)
            if cs <= 100000:  # (the oracle takes dense arrays: small caches only)
                w_o = dw.cpu().numpy().copy()
                lo = np.where(live, loc, 0).astype(np.int32)[live]
                O.cache_backward_rowwise_adagrad_approx(grad, lo, rowidx[live], 0.1, 1e-4, st_o, w_o)
                assert_close(outs[0][1].cpu().numpy(), st_o, "adagrad state", rtol=5e-5, atol_scale=1e-5)
                assert_close(got.cpu().numpy(), w_o, "adagrad rows", rtol=1e-4, atol_scale=1e-5)


def test_rowwise_adagrad_where_the_reference_is_deterministic():
    """Second checker for a12's row-wise Adagrad (round-5 verdict): where the reference's kernel has ONE possible result -- every
    cache row hit from at most one segment (bag), possibly several times inside it -- the product's two kernels (atomic and
    sorted), the oracle and the independent float64 restatement of cu:1735-1795 agree on state and weights."""
    import tt_embeddings as E
    from util import rowwise_adagrad_segments_f64

    rs = np.random.RandomState(5)
    cs, B, D = 3000, 256, 64
    rows_of_bag = rs.permutation(cs)[:B * 6].reshape(B, 6)  # six cache rows per bag, no row in two bags ...
    loc, rowidx = [], []
    for b in range(B):
        pick = rs.choice(6, size=rs.randint(0, 12))             # ... some of them hit several times inside the bag
        loc += [int(rows_of_bag[b][j]) for j in pick]
        rowidx += [b] * len(pick)
    loc, rowidx = np.array(loc, dtype=np.int32), np.array(rowidx, dtype=np.int64)
    n = loc.size
    w = rs.randn(cs, D).astype(np.float32)
    grad = ((rs.rand(B, D) - 0.5) * 0.2).astype(np.float32)
    st0 = (rs.rand(cs) * 0.01).astype(np.float32)
    st64, w64 = rowwise_adagrad_segments_f64(grad, loc, rowidx, 0.1, 1e-4, st0, w)
    st_o, w_o = st0.copy(), w.copy()
    O.cache_backward_rowwise_adagrad_approx(grad, loc, rowidx, 0.1, 1e-4, st_o, w_o)
    assert_close(st_o, st64, "oracle state vs float64 restatement")
    assert_close(w_o, w64, "oracle weights vs float64 restatement")
    for det in (False, True):
        dst, dw = t(st0), t(w)
        E.cache_backward_rowwise_adagrad_approx(n, t(grad), t(loc), t(rowidx), 0.1, 1e-4, dst, dw, deterministic=det)
        assert_close(dst.cpu().numpy(), st64, f"state (deterministic={det}) vs float64 restatement")
        assert_close(dw.cpu().numpy(), w64, f"weights (deterministic={det}) vs float64 restatement")


@pytest.mark.parametrize("tables,p,B,pf,std,H", [
    (1, [20, 22, 25], 300, 10, 3, 1 << 19),     # one launch: rows by binary search over the offsets, fused update
    (1, [20, 22, 25], 400, 8, 4, 0),            # one launch, no frequency table
    (1, [200, 220, 250], 512, 20, 0, 1 << 20),  # the benchmark shape
    (3, [20, 22, 25], 200, 6, 2, 0),            # several tables: separate launches
    (1, [20, 22, 25], 60, 5, 2, 1 << 16),       # nnz <= 1024: separate launches
    (1, [300, 22, 25], 300, 10, 2, 1 << 19),    # a two-pass sort: separate launches
    (1, [20, 22, 25], 5000, 2, 1, 1 << 20),     # more bags than the LDS copy of the offsets holds
    (12, [300, 22, 25], 150, 8, 2, 0),          # 3600 slice ids in core 0: table groups (one table each), wide digit per group
    (40, [60, 22, 25], 40, 6, 2, 0),            # more tables than groups: two tables per group
    (9, [400, 300, 25], 700, 12, 3, 0),         # ~75k lookups, 3600 / 2700 slice ids, several work-groups per group
])
def test_lookup_prologue_equals_separate_calls(tables, p, B, pf, std, H):
    """ttx_lookup_prologue == update_cache_state + preprocess_indices_sync(warmup) + make_plan:
    rowidx / tableidx and the hash table bit-exact (keys whose home slot is contended are only
    compared as a multiset), and the forward output through either plan identical."""
    import tt_embeddings as E

    q, r = [2, 2, 2], [1, 4, 4, 1]
    E_ = int(np.prod(np.array(p, dtype=np.int64)))
    idx, off = G.make_bags(17 + B, B, E_, pf, std, tables)
    off[1:3] = off[1]  # an empty bag near the front (idx beyond stays valid: offsets only move boundaries)
    if tables >= 9:
        off[B + 1:2 * B + 1] = off[B]  # ... and a table without any lookup (its neighbour's first bag takes them)
    cores = [t(c) for c in G.make_cores(3, tables, p, q, r, "signed")]
    ix, of = t(idx), t(off)
    empty64, empty32 = torch.empty(0, dtype=torch.int64, device=DEV), torch.empty(0, dtype=torch.int32, device=DEV)

    def table():
        return (torch.full((H,), -1, dtype=torch.int64, device=DEV), torch.zeros(H, dtype=torch.int64, device=DEV)) if H else (None, None)

    ht1, fr1 = table()
    row1, tab1, plan1 = E.lookup_prologue(ix, of, tables, p, q, r, ht1, fr1)
    ht2, fr2 = table()
    if H:
        E.update_cache_state(ix, ht2, fr2)
    _, row2, tab2, ntt, _ = E.preprocess_indices_sync(ix, of, tables, True, empty64, empty32)
    plan2 = E.make_plan(tables, p, q, r, idx.size, ix, tab2, row2)
    assert ntt == idx.size
    assert torch.equal(row1, row2) and torch.equal(tab1, tab2)
    # oracle for the rows as well
    orow, otab = O.rowidx_from_offsets(off, tables)
    assert np.array_equal(row1.cpu().numpy(), orow) and np.array_equal(tab1.cpu().numpy(), otab)
    if H:  # tables are sparse enough that no key runs out of probes: the stored (key, count) sets must agree
        assert int(fr1.sum()) == int(fr2.sum()) == idx.size
        k1, k2 = ht1.cpu().numpy(), ht2.cpu().numpy()
        f1, f2 = fr1.cpu().numpy(), fr2.cpu().numpy()
        assert sorted(zip(k1[k1 >= 0].tolist(), f1[k1 >= 0].tolist())) == sorted(zip(k2[k2 >= 0].tolist(), f2[k2 >= 0].tolist()))
    Bq, D = B, int(np.prod(q))
    Lt = torch.tensor([int(np.prod(p[i + 1:])) for i in range(3)], dtype=torch.int64, device=DEV)
    o1 = E.tt_forward(1000, tables, Bq, D, p, q, r, Lt, idx.size, ix, row1, tab1, cores, plan=plan1)
    o2 = E.tt_forward(1000, tables, Bq, D, p, q, r, Lt, idx.size, ix, row2, tab2, cores, plan=plan2)
    assert torch.equal(o1, o2)
    g = O.make_geom(tables, p, q, r)
    ref = O.tt_forward(g, Bq, D, idx, orow, otab, [c.cpu().numpy() for c in cores])
    assert_close(o1.cpu().numpy(), ref, "prologue plan forward vs oracle")
    # ... and the backward through either plan (chunk list, slice offsets of every core)
    d_out = t(G.make_grad(19, tables, Bq, D))
    g1 = E.tt_dense_backward(1000, D, p, q, r, Lt, idx.size, ix, row1, tab1, d_out, cores, plan=plan1)
    g2 = E.tt_dense_backward(1000, D, p, q, r, Lt, idx.size, ix, row2, tab2, d_out, cores, plan=plan2)
    gref = O.tt_backward(g, O.OPTIM_DENSE, Bq, D, 0, 0, idx, orow, otab, d_out.cpu().numpy(), [c.cpu().numpy() for c in cores])
    for k in range(3):
        assert_close(g1[k].cpu().numpy(), g2[k].cpu().numpy(), f"prologue plan backward grad{k} vs make_plan's")
        assert_close(g1[k].cpu().numpy(), gref[k], f"prologue plan backward grad{k} vs oracle")


@pytest.mark.parametrize("p", [[20, 22, 25], [300, 29, 31], [2100, 3, 2]])  # 8-bit slice ids / wide digit / two passes
@pytest.mark.parametrize("n_live", [0, 1, 777, 3000])
def test_device_side_counts_equal_exact_sizes(n_live, p):
    """include/ttx.h 'device-side counts': a plan built by ttx_plan_build_n for nnz_dev = n_live out of an
    upper bound nnz, and the cache kernels driven by skip_dev, give what the exact-size calls give."""
    import ctypes as C

    import tt_embeddings as E

    L = E.lib()
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
    L.ttx_plan_build_n.argtypes = [C.POINTER(E._Geom), i64, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_cache_forward_n.argtypes = [i32, i64, vp, vp, vp, i32, vp, vp, vp]
    L.ttx_cache_backward_sgd_n.argtypes = [i64, vp, i32, vp, vp, vp, C.c_float, vp, vp]
    q, r = [4, 4, 4], [1, 16, 16, 1]
    E_, D, B, nnz = p[0] * p[1] * p[2], 64, 150, 3000
    rs = np.random.RandomState(n_live)
    idx = t(rs.randint(0, E_, size=nnz).astype(np.int64))
    rowidx = t(np.sort(rs.randint(0, B, size=nnz)).astype(np.int64))
    tableidx = torch.zeros(nnz, dtype=torch.int64, device=DEV)
    cores = [t(c) for c in G.make_cores(4, 1, p, q, r[1:-1], "signed")]
    Lt = torch.tensor([p[1] * p[2], p[2], 1], dtype=torch.int64, device=DEV)
    g = E._geom(1, p, q, r)
    st = torch.cuda.current_stream().cuda_stream
    n_dev = torch.tensor([n_live], dtype=torch.int32, device=DEV)
    # forward through a plan built with the device-side count
    pb = L.ttx_plan_bytes(C.byref(g), nnz)
    buf = torch.empty(pb, dtype=torch.uint8, device=DEV)
    assert L.ttx_plan_build_n(C.byref(g), nnz, n_dev.data_ptr(), idx.data_ptr(), tableidx.data_ptr(), rowidx.data_ptr(),
                              buf.data_ptr(), pb, st) == 0, L.ttx_last_error()
    plan = E.Plan(buf, nnz, None)
    out_n = E.tt_forward(1000, 1, B, D, p, q, r, Lt, nnz, idx, rowidx, tableidx, cores, plan=plan)
    out_x = E.tt_forward(1000, 1, B, D, p, q, r, Lt, n_live, idx, rowidx, tableidx, cores)
    assert torch.equal(out_n, out_x)
    # fused SGD through the same plan
    d_out = t(G.make_grad(5, 1, B, D))
    ca, cb = [c.clone() for c in cores], [c.clone() for c in cores]
    E.tt_sgd_backward(1000, D, 0.1, p, q, r, Lt, nnz, idx, rowidx, tableidx, d_out, ca, plan=plan)
    E.tt_sgd_backward(1000, D, 0.1, p, q, r, Lt, n_live, idx, rowidx, tableidx, d_out, cb)
    for a, b in zip(ca, cb):
        assert torch.equal(a, b)
    # cache gather / SGD scatter on the entries behind the split point
    rows = 500
    loc = t(rs.randint(0, rows, size=nnz).astype(np.int32))
    w = torch.rand(rows, D, device=DEV)
    o1, o2 = torch.zeros(B, D, device=DEV), torch.zeros(B, D, device=DEV)
    assert L.ttx_cache_forward_n(B, nnz, n_dev.data_ptr(), loc.data_ptr(), rowidx.data_ptr(), D, w.data_ptr(),
                                 o1.data_ptr(), st) == 0
    if nnz - n_live:
        E.cache_forward(B, nnz - n_live, loc[n_live:], rowidx[n_live:], w, o2)
    assert torch.equal(o1, o2)
    w1, w2 = w.clone(), w.clone()
    gr = t(G.make_grad(6, 1, B, D)[0])
    assert L.ttx_cache_backward_sgd_n(nnz, n_dev.data_ptr(), D, gr.data_ptr(), loc.data_ptr(), rowidx.data_ptr(), 0.1,
                                      w1.data_ptr(), st) == 0
    E.cache_backward_sgd(nnz - n_live, gr, loc[n_live:], rowidx[n_live:], 0.1, w2)
    assert_close(w1.cpu().numpy(), w2.cpu().numpy(), "cache SGD scatter behind a device-side split point")


def table_with_cached_keys_behind_emptied_slots(seed=17, H=1 << 14, E_=1 << 20):
    """-> keys, freq, state (a live table), and three key lists: 200 cached keys at their SECOND probe behind an empty first one
    (what cache_populate leaves when it evicts the key that sat in front), 200 cached at their first probe, 1500 new keys whose
    probes stay clear of all of them (a new key racing a re-insert for an empty slot is decided by hardware order, here as in
    the reference), and the generator"""
    rs = np.random.RandomState(seed)
    keys = np.full(H, -1, dtype=np.int64)
    freq = np.zeros(H, dtype=np.int64)
    state = np.full(H, -1, dtype=np.int32)
    behind, front = [], []
    cand = rs.permutation(E_)
    loc = 0
    for k in cand:
        h = O.hash64(int(k), H)
        if keys[h] != -1 or keys[(h + 1) % H] != -1 or keys[(h + 2) % H] != -1 or keys[(h - 1) % H] != -1:
            continue  # (keep the probe sequences of the constructed keys apart)
        if len(behind) < 200:
            keys[(h + 1) % H], freq[(h + 1) % H], state[(h + 1) % H] = k, 50, loc  # cached at the 2nd probe, 1st probe empty
            behind.append(int(k))
        elif len(front) < 200:
            keys[h], freq[h], state[h] = k, 50, loc  # cached at the 1st probe
            front.append(int(k))
        else:
            break
        loc += 1
    taken = np.flatnonzero(keys != -1)
    near = np.zeros(H, dtype=bool)
    for d in range(-3, 4):
        near[(taken + d) % H] = True
    new = [int(k) for k in cand[-6000:] if not near[O.hash64(int(k), H)]][:1500]
    assert len(new) == 1500
    return keys, freq, state, behind, front, new, rs


@pytest.mark.gpu
def test_fused_update_and_lookup_sees_the_batchs_own_inserts():
    """Round 5.  The reference counts a batch into the table in one launch and looks it up in the next (cu:1077-1113, 1356-1375),
    so a look-up sees every insert of ITS batch.  That decides hit or miss for a cached key that sits at its 2nd / 3rd probe
    behind a slot populate emptied: the batch's count re-inserts the key into the empty slot, the look-up finds that copy
    (cache_state -1) and the key is a TT lookup.  Here both happen in one launch (rowidx_update_kernel); a plain find raced with
    the other copies' inserts -- hit or miss by timing.  Constructed: 200 cached keys, every one at its second probe behind an
    empty slot, each looked up ~150 times across many waves and work-groups, next to keys cached at their first probe and new
    keys; partition, split point and table against the oracle run in the reference's order (update, then look-up), ten times."""
    import tt_embeddings as E

    H = 1 << 14
    keys, freq, state, behind, front, new, rs = table_with_cached_keys_behind_emptied_slots(17, H)
    for rep in range(10):
        idx = np.concatenate([rs.choice(behind, 30000), rs.choice(front, 20000), rs.choice(new, 10000)]).astype(np.int64)
        rs.shuffle(idx)
        n, B = idx.size, 512
        off = np.concatenate([[0], np.cumsum(rs.multinomial(n, np.ones(B) / B))]).astype(np.int64)
        ok, of = keys.copy(), freq.copy()
        O.update_cache_state(idx, ok, of)
        exp = O.preprocess_indices(idx, off, 1, False, ok, state)
        dk, df = t(keys), t(freq)
        got = E.preprocess_indices_sync(t(idx), t(off), 1, False, dk, t(state), df)
        assert exp[3] == 40000, "the oracle, in the reference's order: every key behind an empty slot has become a TT lookup"
        assert got[3] == exp[3], f"split point {got[3]} (one launch) vs {exp[3]} (update, then look-up)"
        assert np.array_equal(got[0].cpu().numpy(), exp[0]) and np.array_equal(got[1].cpu().numpy(), exp[1])
        assert np.array_equal(got[4].cpu().numpy()[exp[3]:], exp[4][exp[3]:])
        # the table: the constructed keys sit where the oracle put them, with its counts (new keys that collide with EACH OTHER
        # may swap slots or drop another one of them, as in the reference)
        gk, gf = dk.cpu().numpy(), df.cpu().numpy()
        con = np.isin(ok, np.array(behind + front, dtype=np.int64))
        assert np.array_equal(gk[con], ok[con]) and np.array_equal(gf[con], of[con])
        assert abs(int(gf.sum()) - int(of.sum())) <= 200  # (which new keys are dropped after 3 probes depends on their order)
