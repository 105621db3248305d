"""Duplicate lookups share their contraction (include/ttx.h "duplicate lookups"; north_star: "duplicate rows share
contraction work"): the map of a batch onto its distinct (table, index) pairs, one contraction per pair, bag
pooling through the map, bag gradients of a pair's occurrences summed in index order before the backward.

Against the CPU oracle (which, like the reference, contracts every lookup on its own) and against the plain HIP
path: forward bit-identical to the plain path, gradients / fused optimizers within the fp32 tolerance, the same
from run to run."""
import ctypes as C

import numpy as np
import pytest
import torch

import gen_inputs as G
import oracle_lib as O
from util import EPS, LR, assert_adagrad_close, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make_case(seed, tables, p, q, r, B, pf, dup_frac, dist="signed"):
    rs = np.random.RandomState(seed)
    E_, D = int(np.prod(np.array(p, dtype=np.int64))), int(np.prod(q))
    lens = rs.randint(0, 2 * pf + 1, size=tables * B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(off[-1])
    idx = rs.randint(0, E_, size=nnz).astype(np.int64)
    hot = rs.randint(0, E_, size=max(3, nnz // 200))  # a small set of rows takes dup_frac of the lookups
    pick = rs.zipf(1.5, size=nnz) % hot.size
    idx = np.where(rs.rand(nnz) < dup_frac, hot[pick], idx).astype(np.int64)
    return dict(tables=tables, T=len(p), p=p, q=q, r=G.pad_ranks(r, len(p)), B=B, D=D, indices=idx, offsets=off,
                cores=G.make_cores(seed + 1, tables, p, q, r, dist), d_out=G.make_grad(seed + 2, tables, B, D))


def run(c, mode, dedup):
    import tt_embeddings as E

    tables, p, q, r, B, D = c["tables"], c["p"], c["q"], c["r"], c["B"], c["D"]
    idx, off = t(c["indices"]), t(c["offsets"])
    Lt = torch.zeros(len(p), dtype=torch.int64, device=DEV)
    _, rowidx, tableidx, _, _ = E.preprocess_indices_sync(idx, off, tables, True, torch.empty(0, dtype=torch.int64, device=DEV),
                                                          torch.empty(0, dtype=torch.int32, device=DEV))
    nnz = idx.numel()
    plan = E.make_plan(tables, p, q, r, nnz, idx, tableidx, rowidx, dedup=dedup)
    assert isinstance(plan, E.DedupPlan) == dedup
    cores = [t(x) for x in c["cores"]]
    res = {"out": E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, idx, rowidx, tableidx, cores, plan=plan).cpu().numpy()}
    d_out = t(c["d_out"])
    if mode == "dense":
        res["grads"] = [g.cpu().numpy() for g in E.tt_dense_backward(1000, D, p, q, r, Lt, nnz, idx, rowidx, tableidx, d_out, cores, plan=plan)]
    elif mode == "sgd":
        E.tt_sgd_backward(1000, D, LR, p, q, r, Lt, nnz, idx, rowidx, tableidx, d_out, cores, plan=plan)
    elif mode == "adagrad":
        state = [torch.zeros_like(x) for x in cores]
        E.tt_adagrad_backward(1000, D, LR, EPS, p, q, r, Lt, nnz, idx, rowidx, tableidx, d_out, state, cores, plan=plan)
        res["state"] = [s.cpu().numpy() for s in state]
    res["cores"] = [x.cpu().numpy() for x in cores]
    if dedup:  # the map itself: distinct pairs, occurrences in index order
        L = E.lib()
        dd = plan.dd.cpu().numpy()
        nu = int(np.frombuffer(dd[:4].tobytes(), dtype=np.int32)[0])
        res["nu"] = nu
    return res


def oracle(c, mode):
    g = O.make_geom(c["tables"], c["p"], c["q"], c["r"])
    rowidx, tableidx = O.rowidx_from_offsets(c["offsets"], c["tables"])
    res = {"out": O.tt_forward(g, c["B"], c["D"], c["indices"], rowidx, tableidx, c["cores"])}
    cores = [x.copy() for x in c["cores"]]
    if mode == "dense":
        res["grads"] = O.tt_backward(g, O.OPTIM_DENSE, c["B"], c["D"], 0, 0, c["indices"], rowidx, tableidx, c["d_out"], cores)
    elif mode == "sgd":
        O.tt_backward(g, O.OPTIM_SGD, c["B"], c["D"], LR, 0, c["indices"], rowidx, tableidx, c["d_out"], cores)
    elif mode == "adagrad":
        res["state"] = [np.zeros_like(x) for x in cores]
        O.tt_backward(g, O.OPTIM_ADAGRAD, c["B"], c["D"], LR, EPS, c["indices"], rowidx, tableidx, c["d_out"], cores, res["state"])
    res["cores"] = cores
    return res


CASES = [
    # (tables, p, q, ranks, B, pooling, duplicate fraction)
    (1, [20, 22, 25], [4, 4, 4], [16, 16], 300, 10, 0.7),     # specialised kernels, ~70 % of the lookups on a few rows
    (1, [200, 220, 250], [4, 4, 4], [32, 32], 512, 10, 0.6),  # the benchmark geometry (24-bit keys: three sort passes)
    (3, [7, 9, 11], [3, 4, 5], [13, 12], 120, 6, 0.5),        # generic kernels, three tables, D % 4 == 0
    (2, [5, 8], [3, 5], [6], 90, 5, 0.8),                     # T = 2, D = 15 (scalar pooling / gradient sums)
    (1, [6, 5, 7, 4], [2, 3, 2, 2], [4, 5, 3], 200, 8, 0.9),  # T = 4, 840 rows: nearly every lookup is a duplicate
    (1, [20, 22, 25], [4, 4, 4], [16, 16], 40, 3, 0.0),       # no duplicates to speak of, tiny batch (nnz < 1024)
    (1, [40, 50, 60], [4, 4, 4], [16, 16], 1400, 11, 0.95),   # ~15k lookups (the map's limit is 16384), three rows hot
    (1, [20, 22, 25], [4, 8, 8], [16, 16], 200, 8, 0.85),     # D = 256: the gradient pre-sum's part sums are 64 KB of LDS
    (2, [9, 8, 7], [4, 8, 10], [5, 6], 100, 6, 0.8),          # D = 320, generic kernels
    (1, [9, 8, 7], [8, 8, 10], [4, 4], 60, 5, 0.8),           # D = 640: ten column blocks per gradient row in the pre-sum
    (1, [20, 22, 25], [4, 4, 4], [16, 16], 64, 130, 0.97),    # three rows take nearly all of ~8k lookups: runs over many slices
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_dedup_vs_oracle_and_plain_path(case):
    tables, p, q, r, B, pf, frac = CASES[case]
    c = make_case(100 + case, tables, p, q, r, B, pf, frac)
    nnz = c["indices"].size
    distinct = np.unique(c["indices"] + np.repeat(np.arange(tables), np.diff(c["offsets"][::B])) * int(np.prod(np.array(p, dtype=np.int64)))).size
    for mode in ("dense", "sgd", "adagrad"):
        got, again, plain, orc = run(c, mode, True), run(c, mode, True), run(c, mode, False), oracle(c, mode)
        assert got["nu"] == distinct, "number of distinct (table, index) pairs"
        if frac >= 0.5:
            assert distinct <= 0.62 * nnz, "the case must hold at least ~40-50 % duplicates"
        assert np.array_equal(got["out"], plain["out"]), "forward must be bit-identical to the plain path"
        assert_close(got["out"], orc["out"], f"case {case} out")
        gref = oracle(c, "dense")["grads"] if mode == "adagrad" else None
        # a hot row's gradient is an fp32 sum over thousands of occurrences, added in an order of its own by either
        # side (the oracle: sequentially): the rounding random walk is ~sqrt(n) ulp
        tol = dict(rtol=5e-5, atol_scale=1e-5) if frac >= 0.9 else {}  # (measured worst: 1.5x the default bound, profiles/r05_tolerances.md)
        for k in range(len(p)):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"case {case} grad{k}", **tol)
                assert np.array_equal(got["grads"][k], again["grads"][k]), "not deterministic"
            elif mode == "sgd":
                assert_close(got["cores"][k], orc["cores"][k], f"case {case} sgd core{k}", **tol)
                assert np.array_equal(got["cores"][k], again["cores"][k]), "not deterministic"
            else:
                assert_close(got["state"][k], orc["state"][k], f"case {case} adagrad state{k}", **(dict(rtol=1e-4, atol_scale=2e-5) if tol else {}))
                assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"case {case} adagrad core{k}")


def test_which_batches_the_map_takes():
    import tt_embeddings as E

    p, q, r = [20, 22, 25], [4, 4, 4], [1, 16, 16, 1]
    idx = t(np.random.RandomState(0).randint(0, 11000, size=20000).astype(np.int64))
    tb = torch.zeros_like(idx)
    # (the slice-wise gradient pre-sum keeps nothing in LDS: any embedding dimension is mapped -- D = 640 was refused before)
    assert isinstance(E.make_plan(1, p, [8, 8, 10], r, 100, idx[:100], tb[:100], None, dedup=True), E.DedupPlan)
    assert E.make_plan(1, p, q, r, 0, idx[:0], tb[:0], None, dedup=True) is None  # empty batch: no plan at all
    # (round 3: batches of any size and key spaces beyond 2^32 ARE mapped -- the multi-work-group key sort)
    assert isinstance(E.make_plan(1, p, q, r, idx.numel(), idx, tb, None, dedup=True), E.DedupPlan)
    big = [70000, 70000, 70000]  # 3.4e14 rows: 64-bit keys
    assert isinstance(E.make_plan(1, big, q, r, 100, idx[:100], tb[:100], None, dedup=True), E.DedupPlan)


LARGE = [
    # (tables, p, q, ranks, B, pooling, duplicate fraction): beyond the single-work-group map's 16384 lookups / 32-bit keys
    (1, [40, 50, 60], [4, 4, 4], [16, 16], 2500, 11, 0.9),      # ~27k lookups, three sort passes
    (5, [20, 22, 25], [4, 4, 4], [16, 16], 900, 10, 0.6),       # five tables, ~45k lookups
    (2, [3000, 4000, 5000], [4, 4, 4], [16, 16], 300, 6, 0.7),  # 1.2e11 rows per table: five sort passes of the 64-bit keys
    (1, [7, 9, 11], [3, 4, 5], [13, 12], 4000, 8, 0.5),         # generic kernels, 693 rows: nearly everything a duplicate
    (2, [20, 22, 25], [4, 4, 4], [16, 16], 4200, 10, 0.7),      # ~84k lookups: the wave-span gather pooling (> 65536 lookups)
    (12, [200, 220, 250], [4, 4, 4], [16, 16], 250, 10, 0.7),   # 3000 slice ids in core 2: the plan of the pairs sorts table
                                                                # groups by themselves (first pair of every table = its offsets)
    (12, [200, 220, 250], [4, 4, 4], [16, 16], 250, 10, 0.0),   # ... the same with hardly any duplicates, some tables' groups thin
]


@pytest.mark.parametrize("case", range(len(LARGE)))
def test_dedup_of_large_batches_vs_oracle_and_plain_path(case):
    """the duplicate map for ANY batch size (multi-work-group stable radix sort of 64-bit keys, run heads, scan): forward
    bit-identical to the plain path, gradients / fused SGD against the oracle, deterministic"""
    tables, p, q, r, B, pf, frac = LARGE[case]
    c = make_case(300 + case, tables, p, q, r, B, pf, frac)
    nnz = c["indices"].size
    assert nnz > 16384 or np.prod(np.array(p, dtype=np.float64)) * tables > 2.0**32
    distinct = np.unique(c["indices"] + np.repeat(np.arange(tables), np.diff(c["offsets"][::B])) * int(np.prod(np.array(p, dtype=np.int64)))).size
    for mode in ("dense", "sgd"):
        got, again, plain, orc = run(c, mode, True), run(c, mode, True), run(c, mode, False), oracle(c, mode)
        assert got["nu"] == distinct, "number of distinct (table, index) pairs"
        assert np.array_equal(got["out"], plain["out"]), "forward must be bit-identical to the plain path"
        assert_close(got["out"], orc["out"], f"large case {case} out")
        tol = dict(rtol=5e-5, atol_scale=1e-5)  # (hot rows: sums over thousands of occurrences in an order of their own; measured worst: 1.0x the default bound)
        for k in range(len(p)):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"large case {case} grad{k}", **tol)
                assert np.array_equal(got["grads"][k], again["grads"][k]), "not deterministic"
            else:
                assert_close(got["cores"][k], orc["cores"][k], f"large case {case} sgd core{k}", **tol)
                assert np.array_equal(got["cores"][k], again["cores"][k]), "not deterministic"


@pytest.mark.parametrize("route", ["native-present", "python-only"])
def test_module_with_dedup_tracks_the_plain_module(route, monkeypatch):
    """TTEmbeddingBag(dedup=True) over a Zipf stream (cfg3's, before the cache is populated): same outputs as the
    plain module bit for bit, same cores after several fused-SGD steps to tolerance; the frequency table counts
    every occurrence."""
    import tt_embeddings_ops as ops

    if route == "python-only":
        monkeypatch.setenv("TTX_NO_NATIVE_NODE", "1")
    p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
    E_, D, B, Lp = 11_000_000, 64, 512, 20
    kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform", device=DEV,
              sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=True, cache_size=1024, hashtbl_size=1 << 16)
    torch.manual_seed(3)
    a = ops.TTEmbeddingBag(dedup=True, **kw)
    b = ops.TTEmbeddingBag(dedup=False, **kw)
    with torch.no_grad():
        for x, y in zip(b.tt_cores, a.tt_cores):
            x.copy_(y)
    rs = np.random.RandomState(4)
    off = t(np.arange(0, B * Lp + 1, Lp, dtype=np.int64))
    grad = t((rs.rand(B, D) * 0.1).astype(np.float32))
    for step in range(4):
        idx = t((rs.zipf(1.2, size=B * Lp).astype(np.int64)) % E_)
        oa, ob = a(idx, off), b(idx, off)
        if step == 0:
            assert torch.equal(oa, ob), "first forward: identical cores -> identical output"
        assert_close(oa.detach().cpu().numpy(), ob.detach().cpu().numpy(), f"step {step} output", rtol=2e-5, atol_scale=4e-6)
        oa.backward(grad)
        ob.backward(grad)
    for k in range(3):
        assert_close(a.tt_cores[k].detach().cpu().numpy(), b.tt_cores[k].detach().cpu().numpy(), f"core{k} after 4 steps",
                     rtol=2e-5, atol_scale=4e-6)
    fa, fb = a.cache_freq.cpu().numpy(), b.cache_freq.cpu().numpy()
    assert int(fa.sum()) == int(fb.sum()) == 4 * B * Lp or abs(int(fa.sum()) - int(fb.sum())) < 8


def test_dedup_auto_follows_the_streams_duplicate_share(monkeypatch):
    """dedup="auto": small batches never share; at large batches the module samples the distinct fraction from the
    shared path's own map and keeps sharing on a skewed stream, drops it on a uniform one -- the outputs equal the plain
    module's bit for bit either way (identical cores: no backward in between)."""
    import tt_embeddings_ops as ops

    monkeypatch.setattr(ops, "_DEDUP_AUTO_MIN_NNZ", 20000)
    monkeypatch.setattr(ops, "_DEDUP_AUTO_PERIOD", 3)
    p, q, r = [200, 220, 250], [4, 4, 4], [16, 16]
    E_, D = 11_000_000, 64
    kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform", device=DEV,
              sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05)
    torch.manual_seed(5)
    a = ops.TTEmbeddingBag(dedup="auto", **kw)
    b = ops.TTEmbeddingBag(**kw)
    with torch.no_grad():
        for x, y in zip(b.tt_cores, a.tt_cores):
            x.copy_(y)
    rs = np.random.RandomState(6)

    def batch(B, Lp, zipf):
        off = t(np.arange(0, B * Lp + 1, Lp, dtype=np.int64))
        raw = rs.zipf(1.2, size=B * Lp) if zipf else rs.randint(0, E_, size=B * Lp)
        return t(raw.astype(np.int64) % E_), off

    with torch.no_grad():
        idx, off = batch(256, 20, True)  # 5120 lookups: below the threshold
        assert torch.equal(a(idx, off), b(idx, off)) and a._dd_auto == [False, 0]
        idx, off = batch(1536, 20, True)  # 30720 lookups, skewed: sampled, sharing stays on
        assert torch.equal(a(idx, off), b(idx, off))
        assert a._dd_auto == [True, 3] and a._dd_auto_last < 0.6
        for _ in range(3):
            idx, off = batch(1536, 20, True)
            assert torch.equal(a(idx, off), b(idx, off))
        assert a._dd_auto == [True, 0]
        idx, off = batch(1536, 20, False)  # uniform stream: the next sample turns sharing off
        assert torch.equal(a(idx, off), b(idx, off))
        assert a._dd_auto == [False, 3] and a._dd_auto_last > 0.95
        idx, off = batch(1536, 20, False)
        assert torch.equal(a(idx, off), b(idx, off)) and a._dd_auto == [False, 2]
    # and a training step through the shared path of an auto module
    a._dd_auto = [False, 0]
    idx, off = batch(1536, 20, True)
    oa, ob = a(idx, off), b(idx, off)
    g = t((rs.rand(1536, D) * 0.1).astype(np.float32))
    oa.backward(g)
    ob.backward(g)
    for k in range(3):
        assert_close(a.tt_cores[k].detach().cpu().numpy(), b.tt_cores[k].detach().cpu().numpy(), f"core{k}", rtol=2e-5, atol_scale=4e-6)


@pytest.mark.parametrize("case", [0, 2, 6])
def test_dedup_with_per_sample_weights(case):
    """nn.EmbeddingBag's per_sample_weights through the shared path (C ABI: ttx_tt_forward_dd / ttx_tt_backward_dd with
    weights): forward bit-identical to the weighted plain path, dense gradients and fused SGD within the fp32 tolerance of it
    -- and of the oracle run on the weighted bag gradient of every lookup (a weight scales a lookup's row going in and its
    share of the bag gradient coming back)."""
    import tt_embeddings as E

    tables, p, q, r, B, pf, frac = CASES[case]
    c = make_case(700 + case, tables, p, q, r, B, pf, frac)
    rs = np.random.RandomState(9)
    nnz = c["indices"].size
    w = t((rs.rand(nnz) * 1.5 + 0.25).astype(np.float32))
    idx, off = t(c["indices"]), t(c["offsets"])
    Lt = torch.zeros(len(p), dtype=torch.int64, device=DEV)
    _, rowidx, tableidx, _, _ = E.preprocess_indices_sync(idx, off, tables, True, torch.empty(0, dtype=torch.int64, device=DEV),
                                                          torch.empty(0, dtype=torch.int32, device=DEV))
    D, rp = c["D"], c["r"]
    d_out = t(c["d_out"])
    res = {}
    for dedup in (True, False):
        plan = E.make_plan(tables, p, q, rp, nnz, idx, tableidx, rowidx, dedup=dedup)
        assert isinstance(plan, E.DedupPlan) == dedup
        cores = [t(x) for x in c["cores"]]
        out = E.tt_forward(1000, tables, B, D, p, q, rp, Lt, nnz, idx, rowidx, tableidx, cores, plan=plan, per_sample_weights=w)
        grads = E.tt_dense_backward(1000, D, p, q, rp, Lt, nnz, idx, rowidx, tableidx, d_out, cores, plan=plan, per_sample_weights=w)
        E.tt_sgd_backward(1000, D, LR, p, q, rp, Lt, nnz, idx, rowidx, tableidx, d_out, cores, plan=plan, per_sample_weights=w)
        res[dedup] = (out.cpu().numpy(), [g.cpu().numpy() for g in grads], [x.cpu().numpy() for x in cores])
    assert np.array_equal(res[True][0], res[False][0]), "weighted forward must be bit-identical to the plain weighted path"
    tol = dict(rtol=2e-5, atol_scale=4e-6)
    for k in range(len(p)):
        assert_close(res[True][1][k], res[False][1][k], f"weighted grad{k}", **tol)
        assert_close(res[True][2][k], res[False][2][k], f"weighted sgd core{k}", **tol)
    # oracle: every lookup a bag of its own (B' = nnz, row n), bag gradient w[n] * d_out[bag(n)]
    g = O.make_geom(tables, p, q, rp)
    ro, tb = O.rowidx_from_offsets(c["offsets"], tables)
    wn = w.cpu().numpy()
    per_lookup = (c["d_out"].reshape(tables, B, D)[tb, ro] * wn[:, None]).astype(np.float32)
    own_rows = np.arange(nnz, dtype=np.int64)
    gref = O.tt_backward(g, O.OPTIM_DENSE, nnz, D, 0, 0, c["indices"], own_rows, np.zeros(nnz, dtype=np.int64) if tables == 1 else tb,
                         per_lookup[None] if tables == 1 else None, [x.copy() for x in c["cores"]]) if tables == 1 else None
    if gref is not None:
        for k in range(len(p)):
            assert_close(res[True][1][k], gref[k], f"weighted grad{k} vs oracle", **tol)
