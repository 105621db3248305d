"""The hash-table / cache path against the REFERENCE's own device code (oracle/_ref/libcacheref.so: the
hashtbl_insert / hashtbl_find templates and the raw-pointer cache kernels of tt_embeddings_cuda.cu, compiled
for gfx950 from the line ranges oracle/Makefile selects, run on this GPU).  Three-way where it applies:
libttx (product, through the C ABI shim) == reference kernels == CPU oracle.

This is what pins the oracle's cache restatement to the reference (the reference has no test for this
path): rows a3, a4, a5, a11, a12 (sgd, dense) and the mark/evict step of a13."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import refdev_lib as R
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module", autouse=True)
def _need_refdev():
    assert R.available(), "oracle/_ref/libcacheref.so is missing: run `make -C oracle refdev` in the build container"


def empty_table(H):
    return np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)


def test_kat_json_is_what_the_reference_code_produces():
    """tests/golden/hashtbl_kat.json (hash values, insert returns, final table, find) regenerated from the
    reference's templates running on the GPU must equal the committed file."""
    kat = json.load(open(os.path.join(HERE, "golden", "hashtbl_kat.json")))
    keys = np.array([int(k) for k in kat["hash64"]], dtype=np.int64)
    for j, H in enumerate(kat["sizes"]):
        got = R.hash64(t(keys), H)
        exp = np.array([kat["hash64"][str(int(k))][j] for k in keys], dtype=np.uint32)
        assert np.array_equal(got, exp), f"hash64 mod {H}"
    ins = kat["insert"]
    H = ins["size"]
    k, f = empty_table(H)
    dk, df = t(k), t(f)
    ret = R.insert_seq(t(np.array(ins["keys"], dtype=np.int64)), dk, df).cpu().numpy()
    assert ret.tolist() == ins["returns"]
    assert dk.cpu().numpy().tolist() == ins["final_keys"]
    assert df.cpu().numpy().tolist() == ins["final_freqs"]
    fk = np.array([int(x) for x in ins["find"]], dtype=np.int64)
    assert R.find(t(fk), dk).cpu().numpy().tolist() == [ins["find"][str(int(x))] for x in fk]
    # and the CPU oracle walks the same path
    ok, of = empty_table(H)
    oret = [O.hashtbl_insert(int(x), 1, ok, of) for x in ins["keys"]]
    assert oret == ins["returns"] and ok.tolist() == ins["final_keys"] and of.tolist() == ins["final_freqs"]


@pytest.mark.parametrize("H,nkeys,n", [(64, 40, 600), (257, 200, 3000), (4096, 5000, 6000), (1 << 16, 10 ** 7, 8000)])
def test_sequential_insert_three_way(H, nkeys, n):
    """keys inserted one after the other (probing, overflow and drops included): the reference's template on the
    GPU, the reference's update_cache_state_kernel launched per key, and the CPU oracle leave the same table."""
    rs = np.random.RandomState(H)
    idx = (rs.randint(0, nkeys, size=n).astype(np.int64) * 7919) % 11_000_000
    ok, of = empty_table(H)
    O.update_cache_state(idx, ok, of)
    k1, f1 = (t(a) for a in empty_table(H))
    R.insert_seq(t(idx), k1, f1)
    assert np.array_equal(k1.cpu().numpy(), ok) and np.array_equal(f1.cpu().numpy(), of)
    k2, f2 = (t(a) for a in empty_table(H))
    R.update_cache_state(t(idx[:1500]), k2, f2, sequential=True)
    ok2, of2 = empty_table(H)
    O.update_cache_state(idx[:1500], ok2, of2)
    assert np.array_equal(k2.cpu().numpy(), ok2) and np.array_equal(f2.cpu().numpy(), of2)


def test_update_cache_state_vs_reference_kernel():
    """a4: the product's frequency update and the reference's update_cache_state_kernel (one thread per index,
    racing) on the same batches.  Where no two new keys can race for a slot both must equal the oracle bit for
    bit; in general the order-free content must agree: the multiset of (key, count) of stored keys whenever
    nothing was dropped."""
    import tt_embeddings as E

    rs = np.random.RandomState(5)
    H = 1 << 22  # (load < 1 %: the chance that any key finds its three slots taken is ~1e-2 over the whole test)
    k_p, f_p = (t(a) for a in empty_table(H))
    k_r, f_r = (t(a) for a in empty_table(H))
    ok, of = empty_table(H)
    for step in range(4):
        idx = (rs.zipf(1.2, size=10240).astype(np.int64)) % 11_000_000 if step % 2 else rs.randint(0, 11_000_000, size=10240).astype(np.int64)
        E.update_cache_state(t(idx), k_p, f_p)
        R.update_cache_state(t(idx), k_r, f_r)
        O.update_cache_state(idx, ok, of)

    def content(k, f):
        k, f = k.cpu().numpy() if torch.is_tensor(k) else k, f.cpu().numpy() if torch.is_tensor(f) else f
        m = k != -1
        order = np.argsort(k[m], kind="stable")
        return k[m][order], f[m][order]

    cp, cr, co = content(k_p, f_p), content(k_r, f_r), content(ok, of)
    assert int(cr[1].sum()) == 4 * 10240 and int(cp[1].sum()) == 4 * 10240, "sparse table: nothing may be dropped"
    for a, b in ((cp, cr), (cp, co)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # slot placement may differ between racing orders, but every stored key sits within MAX_PROBES of its home
    for k in (k_p, k_r):
        kk = k.cpu().numpy()
        slots = np.flatnonzero(kk != -1)
        home = R.hash64(t(kk[slots]), H).astype(np.int64)
        assert ((slots - home) % H < 3).all()


@pytest.mark.parametrize("tables,B,pf", [(1, 512, 20), (3, 70, 5), (26, 64, 3), (1, 1, 1), (2, 33, 0)])
def test_compute_rowidx_vs_reference_kernel(tables, B, pf):
    """a3: bag rows / table ids of ragged (and empty) bags"""
    import tt_embeddings as E

    rs = np.random.RandomState(B)
    lens = rs.randint(0, 2 * pf + 1, size=tables * B) if pf else np.zeros(tables * B, dtype=np.int64)
    if pf == 0:
        lens[-1] = 3
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(off[-1])
    idx = t(rs.randint(0, 1000, size=nnz).astype(np.int64))
    rr, rt = R.compute_rowidx(t(off), tables, nnz)
    got = E.preprocess_indices_sync(idx, t(off), tables, True, torch.empty(0, dtype=torch.int64, device=DEV),
                                    torch.empty(0, dtype=torch.int32, device=DEV))
    assert torch.equal(got[1], rr) and torch.equal(got[2], rt)
    orow, otab = O.rowidx_from_offsets(off, tables)
    assert np.array_equal(rr.cpu().numpy(), orow) and np.array_equal(rt.cpu().numpy(), otab)


def _live_table(rs, H, E_, cache_size, batches=5, n=6000, a=1.2):
    keys, freq = empty_table(H)
    for _ in range(batches):
        O.update_cache_state((rs.zipf(a, size=n).astype(np.int64)) % E_, keys, freq)
    state = np.full(H, -1, dtype=np.int32)
    order = np.argsort(-freq, kind="stable")  # stable descending: ties keep ascending slot order
    rows = 0
    for s in order:
        if keys[s] != -1 and rows < cache_size:
            state[s] = rows
            rows += 1
    return keys, freq, state


@pytest.mark.parametrize("H,nnz", [(4096, 3000), (1 << 16, 10240), (1 << 20, 70000)])
def test_lookup_and_partition_vs_reference_kernel(H, nnz):
    """a5 + a6: is_tt / cache_location from the reference's cache_lookup_kernel (its hashtbl_find quirk
    included) define the partition the reference's CUB call performs: selected items in order at the front,
    rejected items REVERSED at the rear (cub::DevicePartition::Flagged).  The product's preprocess must produce
    exactly that from the same inputs; the oracle too."""
    import tt_embeddings as E

    rs = np.random.RandomState(nnz)
    E_ = 11_000_000
    keys, freq, state = _live_table(rs, H, E_, cache_size=H // 8)
    B = 512
    lens = rs.multinomial(nnz, np.ones(B) / B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = (rs.zipf(1.2, size=nnz).astype(np.int64)) % E_
    d_idx, d_keys, d_state = t(idx), t(keys), t(state)
    is_tt, loc = R.cache_lookup(d_idx, d_keys, d_state)
    rrow, _ = R.compute_rowidx(t(off), 1, nnz)
    is_tt_h, loc_h, row_h = is_tt.cpu().numpy(), loc.cpu().numpy(), rrow.cpu().numpy()
    sel, rej = np.flatnonzero(is_tt_h), np.flatnonzero(~is_tt_h)[::-1]
    order = np.concatenate([sel, rej])
    got = E.preprocess_indices_sync(d_idx, t(off), 1, False, d_keys, d_state)
    assert got[3] == sel.size
    assert np.array_equal(got[0].cpu().numpy(), idx[order]), "partitioned colidx"
    assert np.array_equal(got[1].cpu().numpy(), row_h[order]), "partitioned rowidx"
    assert np.array_equal(got[4].cpu().numpy()[sel.size:], loc_h[rej]), "cache locations of the cached entries"
    exp = O.preprocess_indices(idx, off, 1, False, keys, state)
    assert exp[3] == sel.size and np.array_equal(exp[0], idx[order]) and np.array_equal(exp[4][sel.size:], loc_h[rej])
    assert 0 < sel.size < nnz, "the case must mix cached and uncached lookups"


@pytest.mark.parametrize("D,nnz,B,cs", [(64, 9000, 512, 4096), (128, 2000, 64, 300), (64, 50000, 4096, 65536), (4, 700, 16, 50)])
def test_cache_gather_and_scatter_vs_reference_kernels(D, nnz, B, cs):
    """a11 / a12: cache_forward, cache_backward_sgd and cache_backward_dense against the reference's kernels on
    the same (rowidx-sorted runs of) cached lookups.  Forward sums are order-free per output element up to fp32
    rounding of the run order; the reference's atomics leave the add order to the hardware -> 1e-5."""
    import tt_embeddings as E

    rs = np.random.RandomState(D + nnz)
    rows = np.sort(rs.randint(0, B, size=nnz)).astype(np.int64)
    loc = (rs.zipf(1.3, size=nnz) % cs).astype(np.int32)
    w = rs.randn(cs, D).astype(np.float32)
    out0 = rs.randn(1, B, D).astype(np.float32)
    ref_out, got_out = t(out0), t(out0)
    R.cache_forward(t(rows), t(loc), t(w), ref_out)
    E.cache_forward(B, nnz, t(loc), t(rows), t(w), got_out)
    assert_close(got_out.cpu().numpy(), ref_out.cpu().numpy(), "cache_forward vs reference kernel")
    o_out = out0.copy()
    O.cache_forward(B, loc, rows, w, o_out[0])
    assert_close(o_out, ref_out.cpu().numpy(), "oracle cache_forward vs reference kernel")
    grad = (rs.rand(B, D) * 0.1).astype(np.float32)
    ref_w, got_w = t(w), t(w)
    R.cache_backward_sgd(t(grad), t(loc), t(rows), 0.1, ref_w)
    E.cache_backward_sgd(nnz, t(grad), t(loc), t(rows), 0.1, got_w)
    # a hot row takes thousands of adds in hardware order in the reference: compare both with the float64 sum
    w64 = w.astype(np.float64)
    np.subtract.at(w64, loc, np.float64(np.float32(0.1)) * grad[rows].astype(np.float64))
    assert_close(ref_w.cpu().numpy(), w64, "reference cache_backward_sgd vs float64", rtol=2e-5, atol_scale=1e-5)
    assert_close(got_w.cpu().numpy(), w64, "cache_backward_sgd vs float64", rtol=2e-5, atol_scale=1e-5)
    ref_g = R.cache_backward_dense(t(grad), t(loc), t(rows), cs)
    got_g = E.cache_backward_dense(nnz, t(grad), t(loc), t(rows), 0.1, t(w))
    g64 = np.zeros((cs, D))
    np.add.at(g64, loc, grad[rows].astype(np.float64))
    assert_close(ref_g.cpu().numpy(), g64, "reference cache_backward_dense vs float64", rtol=2e-5, atol_scale=1e-5)
    assert_close(got_g.cpu().numpy(), g64, "cache_backward_dense vs float64", rtol=2e-5, atol_scale=1e-5)


@pytest.mark.parametrize("H,cache_size", [(4096, 64), (1 << 16, 1000), (512, 300), (1 << 20, 262144)])
def test_populate_mark_and_evict_vs_reference_kernel(H, cache_size):
    """a13, middle step: given the table sorted by descending frequency (stable: cub::DeviceRadixSort keeps
    equal keys in input order), the reference's mark_popular_colidx_kernel assigns cache rows to the top
    cache_size keys and evicts the rest.  The product's cache_populate (own radix sort + mark) must leave the
    same hashtbl / cache_freq / cache_state."""
    import gen_inputs as G
    import tt_embeddings as E

    rs = np.random.RandomState(cache_size)
    p, q, r = [20, 22, 25], [4, 4, 4], [1, 16, 16, 1]
    E_ = 11000 if H <= (1 << 16) else 11_000_000
    if E_ > 11000:
        p = [200, 220, 250]
    keys, freq = empty_table(H)
    nb = 6 if H <= (1 << 16) else 60
    for _ in range(nb):
        O.update_cache_state((rs.zipf(1.2, size=50000 if nb > 6 else 4000).astype(np.int64)) % E_, keys, freq)
    order = np.argsort(-freq, kind="stable")
    sorted_keys = keys[order].copy()
    rk, rf, rstate = t(keys), t(freq), t(np.full(H, -1, dtype=np.int32))
    R.mark_popular(cache_size, t(sorted_keys), rk, rf, rstate)
    cores = [t(c) for c in G.make_cores(3, 1, p, q, r)]
    pk, pf_, pstate = t(keys), t(freq), t(np.full(H, -1, dtype=np.int32))
    pw = torch.zeros(cache_size, 64, device=DEV)
    E.cache_populate(E_, p, q, r, cores, torch.zeros(3, dtype=torch.int64, device=DEV), pk, pf_, pstate, pw)
    assert torch.equal(pk, rk), "hashtbl after eviction"
    assert torch.equal(pf_, rf), "cache_freq after eviction"
    assert torch.equal(pstate, rstate), "cache_state"
    # the decompressed rows are the TT rows of the top keys, in rank order (prefetch_cached_weights_cuda)
    top = sorted_keys[:cache_size].copy()
    n_rows = int((top != -1).sum())
    sub = np.flatnonzero(top != -1)[:: max(1, n_rows // 512)]
    rows = O.tt_rows(O.make_geom(1, p, q, r), 64, top[sub], None, [c.cpu().numpy() for c in cores])
    assert_close(pw.cpu().numpy()[sub], rows, "decompressed cache rows (sub-sample)")


def test_second_populate_differs_from_reference_only_on_evicted_slots():
    """f3 / deliberate deviation: the reference's mark_popular_colidx_kernel leaves
    cache_state[slot] untouched when it evicts a key, so after a SECOND populate an evicted slot still carries the
    cache row it had before -- the next key inserted there is served (and updates) another index's row.  The
    product (and the oracle) reset it to -1.  Everything else of the second populate must equal the reference
    kernel's result: hashtbl, cache_freq, and cache_state on every slot that was not evicted."""
    import gen_inputs as G
    import tt_embeddings as E

    rs = np.random.RandomState(17)
    H, cs, E_ = 1 << 14, 1500, 11000
    p, q, r = [20, 22, 25], [4, 4, 4], [1, 16, 16, 1]
    cores = [t(c) for c in G.make_cores(3, 1, p, q, r)]
    Lt = torch.zeros(3, dtype=torch.int64, device=DEV)
    keys, freq = empty_table(H)
    for _ in range(6):
        O.update_cache_state((rs.zipf(1.3, size=4000).astype(np.int64)) % E_, keys, freq)
    dk, df, ds, dw = t(keys), t(freq), t(np.full(H, -1, dtype=np.int32)), torch.zeros(cs, 64, device=DEV)
    E.cache_populate(E_, p, q, r, cores, Lt, dk, df, ds, dw)           # first populate (== reference, tested above)
    keys1, freq1, state1 = dk.cpu().numpy(), df.cpu().numpy(), ds.cpu().numpy()
    assert int((state1 >= 0).sum()) == cs
    # the stream moves on: a different popularity profile, inserted sequentially (one defined table state)
    # Only lookups that cannot leave a key in TWO slots: keys sitting on their home slot, and new keys whose home
    # slot is free.  (A key stored behind an evicted slot is re-inserted in front of its old copy -- in the reference
    # too -- and mark_popular's outcome for such twins depends on which thread runs first, in either build.)
    present = set(int(k) for k in keys1[keys1 != -1])
    keep = []
    for k in ((rs.zipf(1.15, size=32000).astype(np.int64) * 7 + 3) % E_).tolist():
        h = O.hash64(k, H)
        if keys1[h] == k:
            keep.append(k)
        elif keys1[h] == -1 and k not in present:
            keep.append(k)
            keys1[h] = k
            present.add(k)
    keys1[freq1 == 0] = -1  # (slots claimed by the filter above: the oracle insert below fills them again)
    O.update_cache_state(np.array(keep, dtype=np.int64), keys1, freq1)
    assert len(keep) > 8000
    order = np.argsort(-freq1, kind="stable")
    rk, rf, rstate = t(keys1), t(freq1), t(state1)
    R.mark_popular(cs, t(keys1[order].copy()), rk, rf, rstate)          # the reference's kernel on the stale state
    pk, pf_, pstate = t(keys1), t(freq1), t(state1)
    E.cache_populate(E_, p, q, r, cores, Lt, pk, pf_, pstate, dw)
    assert torch.equal(pk, rk) and torch.equal(pf_, rf), "table contents after the second populate"
    got, ref = pstate.cpu().numpy(), rstate.cpu().numpy()
    evicted = (keys1 != -1) & (rk.cpu().numpy() == -1)
    assert evicted.sum() > 100, "the case must evict keys"
    assert np.array_equal(got[~evicted], ref[~evicted]), "cache_state of the slots that were not evicted"
    assert (got[evicted] == -1).all(), "evicted slots must not keep a cache row"
    stale = int((ref[evicted] != -1).sum())
    assert stale > 0, "the reference leaves stale rows on evicted slots here (otherwise the case shows nothing)"
    ok, of, ost = keys1.copy(), freq1.copy(), state1.copy()
    O.cache_populate(O.make_geom(1, p, q, r), [c.cpu().numpy() for c in cores], ok, of, ost, np.zeros((cs, 64), dtype=np.float32))
    assert np.array_equal(ost, got) and np.array_equal(ok, pk.cpu().numpy())
    # ... and with the fix switched off PER CALL (round 6: ttx_cache_populate_f(flags = TTX_POPULATE_REFERENCE_EXACT), the module's
    # ctor keyword `reference_exact_populate`; no process-wide switch) product, reference kernel and oracle agree on EVERY slot
    O.set_reference_exact(1)
    try:
        xk, xf, xstate = t(keys1), t(freq1), t(state1)
        E.cache_populate(E_, p, q, r, cores, Lt, xk, xf, xstate, dw, reference_exact=True)
        assert torch.equal(xk, rk) and torch.equal(xf, rf) and torch.equal(xstate, rstate), "reference-exact second populate"
        # the flag is the call's, not the library's: the next plain call is the default again
        yk, yf, ystate = t(keys1), t(freq1), t(state1)
        E.cache_populate(E_, p, q, r, cores, Lt, yk, yf, ystate, dw)
        assert torch.equal(ystate, pstate)
        # and through the module: the ctor keyword reaches the call
        import tt_embeddings_ops as ops
        for exact, want in ((True, rstate), (False, pstate)):
            m = ops.TTEmbeddingBag(E_, 64, r[1:-1], p, q, use_cache=True, cache_size=cs, hashtbl_size=H, weight_dist="uniform",
                                   device=DEV, reference_exact_populate=exact)
            with torch.no_grad():
                for dst, src in zip(m.tt_cores, cores):
                    dst.copy_(src.reshape(dst.shape))
                m.hashtbl.copy_(t(keys1)); m.cache_freq.copy_(t(freq1)); m.cache_state.copy_(t(state1))
            m.cache_populate()
            assert torch.equal(m.cache_state, want) and torch.equal(m.hashtbl, rk)
        ok, of, ost = keys1.copy(), freq1.copy(), state1.copy()
        O.cache_populate(O.make_geom(1, p, q, r), [c.cpu().numpy() for c in cores], ok, of, ost, np.zeros((cs, 64), dtype=np.float32))
        assert np.array_equal(ost, ref) and np.array_equal(ok, rk.cpu().numpy()) and np.array_equal(of, rf.cpu().numpy())
    finally:
        O.set_reference_exact(0)


def test_lookup_after_the_batchs_own_update_vs_reference_kernels():
    """Round 5: the reference counts a batch into the table (update_cache_state_kernel) and looks it up in the NEXT launch
    (cache_lookup_kernel); the product does both in one.  On a table with cached keys behind emptied slots -- where the batch's
    own re-insert decides hit or miss -- the reference's two kernels, run here one after the other, and the product's one launch
    give the same is_tt / cache locations / partition and leave the same counts on the constructed keys."""
    import tt_embeddings as E
    from test_cache_gpu import table_with_cached_keys_behind_emptied_slots

    H = 1 << 14
    keys, freq, state, behind, front, new, rs = table_with_cached_keys_behind_emptied_slots(23, H)
    for rep in range(3):
        idx = np.concatenate([rs.choice(behind, 30000), rs.choice(front, 20000), rs.choice(new, 10000)]).astype(np.int64)
        rs.shuffle(idx)
        n, B = idx.size, 512
        off = np.concatenate([[0], np.cumsum(rs.multinomial(n, np.ones(B) / B))]).astype(np.int64)
        rk, rf = t(keys), t(freq)
        R.update_cache_state(t(idx), rk, rf)
        is_tt, loc = R.cache_lookup(t(idx), rk, t(state))
        is_tt_h, loc_h = is_tt.cpu().numpy(), loc.cpu().numpy()
        assert int(is_tt_h.sum()) == 40000, "the reference: every key behind an emptied slot is a TT lookup after its batch's update"
        sel, rej = np.flatnonzero(is_tt_h), np.flatnonzero(~is_tt_h)[::-1]
        pk, pf = t(keys), t(freq)
        got = E.preprocess_indices_sync(t(idx), t(off), 1, False, pk, t(state), pf)
        assert got[3] == sel.size
        assert np.array_equal(got[0].cpu().numpy(), idx[np.concatenate([sel, rej])]), "partitioned colidx"
        assert np.array_equal(got[4].cpu().numpy()[sel.size:], loc_h[rej]), "cache locations of the cached entries"
        con = np.isin(rk.cpu().numpy(), np.array(behind + front, dtype=np.int64))
        assert np.array_equal(pk.cpu().numpy()[con], rk.cpu().numpy()[con]) and np.array_equal(pf.cpu().numpy()[con], rf.cpu().numpy()[con])
