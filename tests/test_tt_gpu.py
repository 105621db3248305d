"""GPU parity of the TT contraction path: libttx (HIP, through the C ABI shim
`tt_embeddings`) vs the CPU oracle and vs the golden vectors generated from the
reference's own Python oracle.  Mirrors the reference's six property tests
(tt_embeddings_test.py:62-525) with fixed seeds."""
import numpy as np
import pytest
import torch

import gen_inputs as G
import oracle_lib as O
from util import LR, EPS, adagrad_expected, assert_adagrad_close, assert_close, sgd_expected

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def run_case(c, mode, plan_shared=False):
    """-> dict(out, grads | cores, state) computed on the GPU"""
    import tt_embeddings as E

    tables, p, q, r, B, D = c["tables"], c["p"], c["q"], c["r"], c["B"], c["D"]
    idx, off = t(c["indices"]), t(c["offsets"])
    Lt = torch.zeros(len(p), dtype=torch.int64, device=dev())
    colidx, rowidx, tableidx, ntt, loc = E.preprocess_indices_sync(
        idx, off, tables, True, torch.empty(0, dtype=torch.int64, device=dev()), torch.empty(0, dtype=torch.int32, device=dev()))
    nnz = idx.numel()
    assert ntt == nnz and loc is None
    cores = [t(x) for x in c["cores"]]
    plan = E.make_plan(tables, p, q, r, nnz, colidx, tableidx, rowidx if (nnz % 2 == 0) else None) if plan_shared else None
    out = E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, colidx, rowidx, tableidx, cores, plan=plan)
    res = {"out": out.cpu().numpy(), "rowidx": rowidx.cpu().numpy(), "tableidx": tableidx.cpu().numpy()}
    d_out = t(c["d_out"])
    if mode == "dense":
        grads = E.tt_dense_backward(1000, D, p, q, r, Lt, nnz, colidx, rowidx, tableidx, d_out, cores, plan=plan)
        res["grads"] = [g.cpu().numpy() for g in grads]
    elif mode == "sgd":
        E.tt_sgd_backward(1000, D, LR, p, q, r, Lt, nnz, colidx, rowidx, tableidx, d_out, cores, plan=plan)
    elif mode == "adagrad":
        state = [torch.zeros_like(x) for x in cores]
        E.tt_adagrad_backward(1000, D, LR, EPS, p, q, r, Lt, nnz, colidx, rowidx, tableidx, d_out, state, cores, plan=plan)
        res["state"] = [s.cpu().numpy() for s in state]
    res["cores"] = [x.cpu().numpy() for x in cores]
    return res


def oracle_case(c, mode):
    g = O.make_geom(c["tables"], c["p"], c["q"], c["r"])
    rowidx, tableidx = O.rowidx_from_offsets(c["offsets"], c["tables"])
    out = O.tt_forward(g, c["B"], c["D"], c["indices"], rowidx, tableidx, c["cores"])
    cores = [x.copy() for x in c["cores"]]
    res = {"out": out, "rowidx": rowidx, "tableidx": tableidx}
    if mode == "dense":
        res["grads"] = O.tt_backward(g, O.OPTIM_DENSE, c["B"], c["D"], 0, 0, c["indices"], rowidx, tableidx, c["d_out"], cores)
    elif mode == "sgd":
        O.tt_backward(g, O.OPTIM_SGD, c["B"], c["D"], LR, 0, c["indices"], rowidx, tableidx, c["d_out"], cores)
    elif mode == "adagrad":
        state = [np.zeros_like(x) for x in cores]
        O.tt_backward(g, O.OPTIM_ADAGRAD, c["B"], c["D"], LR, EPS, c["indices"], rowidx, tableidx, c["d_out"], cores, state)
        res["state"] = state
    res["cores"] = cores
    return res


def test_rowidx(small_cases):
    for name, c in small_cases.items():
        got = run_case(c, "fwd")
        r, tb = O.rowidx_from_offsets(c["offsets"], c["tables"])
        assert np.array_equal(got["rowidx"], r) and np.array_equal(got["tableidx"], tb), name


@pytest.mark.parametrize("shared", [False, True])
def test_forward_golden(small_cases, shared):
    """tt_embeddings_test.py:62-107 (test_forward) and :343-425 (table batched)"""
    for name, c in small_cases.items():
        got = run_case(c, "fwd", shared)
        assert_close(got["out"], c["out"], f"{name} out vs golden")
        assert_close(got["out"], oracle_case(c, "fwd")["out"], f"{name} out vs oracle")


def test_backward_dense_golden(small_cases):
    """tt_embeddings_test.py:116-174, :435-525"""
    for name, c in small_cases.items():
        got = run_case(c, "dense", True)
        orc = oracle_case(c, "dense")
        for k in range(c["T"]):
            assert_close(got["grads"][k], c["grads"][k], f"{name} grad{k} vs golden")
            assert_close(got["grads"][k], orc["grads"][k], f"{name} grad{k} vs oracle")
            assert np.array_equal(got["cores"][k], c["cores"][k]), "dense backward must not touch the cores"


def test_backward_sgd_golden(small_cases):
    """tt_embeddings_test.py:183-246"""
    for name, c in small_cases.items():
        got = run_case(c, "sgd")
        exp = sgd_expected(c["cores"], c["grads"])
        orc = oracle_case(c, "sgd")
        for k in range(c["T"]):
            assert_close(got["cores"][k], exp[k], f"{name} sgd core{k} vs golden")
            assert_close(got["cores"][k], orc["cores"][k], f"{name} sgd core{k} vs oracle")


def test_backward_adagrad_golden(small_cases):
    """tt_embeddings_test.py:255-333"""
    for name, c in small_cases.items():
        got = run_case(c, "adagrad", True)
        exp, st = adagrad_expected(c["cores"], c["grads"])
        orc = oracle_case(c, "adagrad")
        for k in range(c["T"]):
            assert_close(got["state"][k], st[k], f"{name} adagrad state{k} vs golden")
            assert_close(got["state"][k], orc["state"][k], f"{name} adagrad state{k} vs oracle")
            assert_adagrad_close(got["cores"][k], exp[k], c["grads"][k], f"{name} adagrad core{k} vs golden")
            assert_adagrad_close(got["cores"][k], orc["cores"][k], c["grads"][k], f"{name} adagrad core{k} vs oracle")


def test_round4_geometries_golden(round4_cases):
    """tests/golden/round4_cases.npz -- what the reference's Python gives for the geometry classes round 4 moved onto new routes:
    q2 = 12 / 16 on the q2 <= 16 templates (padded, and exact at r = 64), four cores with a merged last factor of 8 / 16 on the
    three-core kernels, two cores at ranks that are not multiples of 4 on the dedicated kernels (q0 = 5, 7, 8 run their part
    lookups in the module: tests/test_module_gpu.py; here they take the generic kernels).  Forward, dense, SGD, Adagrad."""
    import tt_embeddings as E

    for name, c in round4_cases.items():
        spec = E.debug_tiles(c["tables"], c["p"], c["q"], c["r"])["MC"] == 0
        assert spec == (not name.startswith(("t3_q5", "t3_q7", "t3_q8"))), f"{name}: route"
        got = run_case(c, "dense", True)
        assert_close(got["out"], c["out"], f"{name} out vs golden")
        for k in range(c["T"]):
            assert_close(got["grads"][k], c["grads"][k], f"{name} grad{k} vs golden")
        got = run_case(c, "sgd")
        for k, e in enumerate(sgd_expected(c["cores"], c["grads"])):
            assert_close(got["cores"][k], e, f"{name} sgd core{k} vs golden")
        got = run_case(c, "adagrad", True)
        exp, st = adagrad_expected(c["cores"], c["grads"])
        for k in range(c["T"]):
            assert_close(got["state"][k], st[k], f"{name} adagrad state{k} vs golden")
            assert_adagrad_close(got["cores"][k], exp[k], c["grads"][k], f"{name} adagrad core{k} vs golden")


def _random_case(seed, T, tables, B, pf, std, dist="uniform"):
    p, q, r = G.test_shape(T)
    r = G.pad_ranks(r, T)
    E_ = int(np.prod(p))
    idx, off = G.make_bags(seed, B, E_, pf, std, tables)
    return dict(tables=tables, T=T, p=p, q=q, r=r, B=B, D=int(np.prod(q)), indices=idx, offsets=off,
                cores=G.make_cores(seed + 1, tables, p, q, r, dist), d_out=G.make_grad(seed + 2, tables, B, int(np.prod(q))))


@pytest.mark.parametrize("T", [2, 3, 4])
@pytest.mark.parametrize("tables", [1, 4])
def test_property_sweep_vs_oracle(T, tables):
    """the hypothesis ranges of the reference tests (batch 200..500, pooling 1..10,
    std 0..20, tables 1..4), fixed seeds, all three backward modes vs the oracle"""
    rs = np.random.RandomState(1000 * T + tables)
    for it in range(3):
        B = int(rs.randint(200, 501))
        pf = int(rs.randint(1, 11))
        std = int(rs.randint(0, 21))
        c = _random_case(int(rs.randint(1 << 30)), T, tables, B, pf, std, "uniform" if it < 2 else "signed")
        for mode in ("dense", "sgd", "adagrad"):
            got = run_case(c, mode, plan_shared=(it % 2 == 0))
            orc = oracle_case(c, mode)
            assert_close(got["out"], orc["out"], f"T{T} tb{tables} it{it} out")
            gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
            for k in range(T):
                if mode == "dense":
                    assert_close(got["grads"][k], orc["grads"][k], f"T{T} tb{tables} it{it} grad{k}")
                elif mode == "sgd":
                    assert_close(got["cores"][k], orc["cores"][k], f"T{T} tb{tables} it{it} {mode} core{k}")
                if mode == "adagrad":
                    assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"T{T} tb{tables} it{it} adagrad core{k}")
                    assert_close(got["state"][k], orc["state"][k], f"T{T} tb{tables} it{it} state{k}")


def test_large_batch_uses_the_global_memory_plan():
    """nnz > 16384 takes the count / scatter launches of the multi-work-group plan; nnz below
    takes the single-launch one.  Both against the oracle (dense grads + fused SGD)."""
    for B, pf, tables in ((700, 9, 3), (500, 8, 4), (2000, 12, 1)):
        c = _random_case(77 + B, 3, tables, B, pf, 2)
        assert (c["indices"].size > 16384) == (B != 500)
        for mode in ("dense", "sgd"):
            got, orc = run_case(c, mode, plan_shared=True), oracle_case(c, mode)
            assert_close(got["out"], orc["out"], f"B{B} out")
            for k in range(3):
                if mode == "dense":
                    assert_close(got["grads"][k], orc["grads"][k], f"B{B} grad{k}")
                else:
                    assert_close(got["cores"][k], orc["cores"][k], f"B{B} sgd core{k}")


@pytest.mark.parametrize("tables,p,B,pf", [
    (1, [300, 290, 310], 300, 10),    # two 8-bit passes per core, bases derived in the scatter pass, finish launch
    (1, [300, 290, 310], 3000, 10),   # 118 wave units of 256
    (2, [300, 29, 310], 5000, 20),    # ~200k lookups: longer wave units, still scanned inside the scatter pass
    (1, [40, 300, 50], 40000, 30),    # ~1.2M lookups: more than 256 units of 4096 -> the scan launch
    (3, [100, 120, 90], 400, 6),      # slice id = table * p + i_t exceeds one digit because of the table
    (2, [20, 600, 15], 350, 7),       # only the pivot core needs a second pass
    (1, [256, 255, 257], 500, 30),    # digit-boundary slice counts, single/multi pass mixed, N < 16384
    (1, [2500, 30, 20], 600, 12),     # 2048 < slices <= 4096: the 12-bit wide digit (round 5: 16-bit per-wave rows)
    (1, [4096, 20, 30], 700, 9),      # ... at its upper edge: 4096 slices, every partial-chunk counter in use
    (2, [1700, 25, 20], 900, 14),     # ... through the table id: 3400 slice ids
    (1, [5000, 30, 20], 600, 12),     # more than 4096 slices in a core: two 8-bit passes (no wide digit)
    (7, [250, 260, 240], 300, 8),     # 7 tables: 1820 slice ids -> the 11-bit wide digit
    (4, [250, 220, 200], 1100, 20),   # the table-batched benchmark geometry: 10-bit wide digit, ~10 work-groups
])
def test_plan_paths_vs_oracle(tables, p, B, pf):
    """every route through the lookup plan (ttx_plan.hip): single launch (all sorts one 8-bit pass,
    N <= 16384) is what the other tests take; here the wide-digit single pass (256 < slices <= 4096,
    up to 96 work-groups), multi-pass sorts, the separate scan launch and the finish launch.  Forward, dense grads and fused SGD against the oracle."""
    q, r = [2, 3, 2], [1, 4, 5, 1]
    E_, D = int(np.prod(np.array(p, dtype=np.int64))), int(np.prod(q))
    idx, off = G.make_bags(5 + B, B, E_, pf, 1, tables)
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(6 + B, tables, p, q, r, "signed"), d_out=G.make_grad(7, tables, B, D))
    for mode in ("dense", "sgd"):
        got, orc = run_case(c, mode, plan_shared=True), oracle_case(c, mode)
        assert_close(got["out"], orc["out"], f"plan {p} out")
        for k in range(3):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"plan {p} grad{k}")
            else:
                assert_close(got["cores"][k], orc["cores"][k], f"plan {p} sgd core{k}")


def test_wide_digit_plan_declines_more_than_2_to_18_full_chunks():
    """Round 6 (advisor): the wide-digit plan's chunk list packs a slice's full-chunk count into 18 bits of an int32 prefix sum;
    with ONE lookup per chunk (the generic kernels' last-resort tile; here the test knob) and more than 262,144 lookups the sum
    overflowed.  Such a batch now takes the multi-pass plan: forward and fused SGD against the oracle at 280k lookups."""
    import tt_embeddings as E

    p, q, r, tables = [300, 29, 310], [2, 3, 2], [1, 4, 5, 1], 2
    E_, D, B = int(np.prod(np.array(p, dtype=np.int64))), int(np.prod(q)), 14000
    idx, off = G.make_bags(91, B, E_, 10, 1, tables)
    assert 262144 < idx.size <= 96 * 4096
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(92, tables, p, q, r, "signed"), d_out=G.make_grad(93, tables, B, D))
    E.set_chunk(1)
    try:
        got = {mode: run_case(c, mode, plan_shared=True) for mode in ("dense", "sgd")}
    finally:
        E.set_chunk(0)
    for mode in ("dense", "sgd"):
        orc = oracle_case(c, mode)
        assert_close(got[mode]["out"], orc["out"], "MC = 1 out")
        for k in range(3):
            if mode == "dense":
                assert_close(got[mode]["grads"][k], orc["grads"][k], f"MC = 1 grad{k}")
            else:
                assert_close(got[mode]["cores"][k], orc["cores"][k], f"MC = 1 sgd core{k}")


@pytest.mark.parametrize("p,tables", [([100, 120, 90], 3), ([2500, 30, 20], 1), ([5000, 30, 20], 1)])  # wide digit (10 / 12 bits) / two passes
@pytest.mark.parametrize("nnz", [4096, 4097, 8191, 12288])
@pytest.mark.parametrize("skew", [False, True])
def test_plan_work_group_boundaries(p, tables, nnz, skew):
    """the 4096-position work-groups of the wide-digit and the multi-pass plan: lookup counts on and
    next to the work-group boundary, uniform and with nearly every lookup in one slice per core
    (one histogram bin takes the whole work-group; a slice's chunk count exceeds its neighbours')."""
    q, r = [2, 3, 2], [1, 4, 5, 1]
    E_, D, B = int(np.prod(np.array(p, dtype=np.int64))), int(np.prod(q)), 64
    rs = np.random.RandomState(nnz + tables)
    sizes = rs.multinomial(nnz, np.ones(tables * B) / (tables * B))  # ragged bags, some empty
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    idx = rs.randint(0, E_, size=nnz).astype(np.int64)
    if skew:
        hot = rs.randint(0, E_)
        idx = np.where(rs.rand(nnz) < 0.9, hot, idx).astype(np.int64)
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(11, tables, p, q, r, "signed"), d_out=G.make_grad(12, tables, B, D))
    # a hot slice's gradient is an fp32 sum of up to 11k signed terms (partial sums ~100) whose order differs
    # from the oracle's sequential one: the rounding random walk is ~sqrt(n) * 6e-8 * 100
    tol = dict(rtol=5e-5, atol_scale=1e-5) if skew else {}
    for mode in ("dense", "sgd"):
        got, orc = run_case(c, mode, plan_shared=True), oracle_case(c, mode)
        assert_close(got["out"], orc["out"], f"boundary {nnz} out")
        for k in range(3):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"boundary {nnz} grad{k}", **tol)
            else:
                assert_close(got["cores"][k], orc["cores"][k], f"boundary {nnz} sgd core{k}", **tol)


@pytest.mark.parametrize("p,q,r", [([3, 2, 4], [4, 4, 4], [1, 16, 16, 1]), ([2, 3, 3], [2, 3, 2], [1, 4, 5, 1]),
                                   ([3, 2], [4, 4], [1, 8, 1])])
def test_hot_slices_are_reduced_by_several_work_groups(p, q, r):
    """a skewed stream in miniature: few slices, thousands of lookups each -> reduce_apply's segment
    work-groups and last-arriver fold (thin cores: > 1024 lookups per slice, pivot: > 64 chunks per slice);
    dense gradients, fused SGD and Adagrad against the oracle, and bit-identical from run to run"""
    E_, D, B = int(np.prod(p)), int(np.prod(q)), 300
    idx, off = G.make_bags(41, B, E_, 25, 3, 1)
    assert idx.size > 6000
    c = dict(tables=1, T=len(p), p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(42, 1, p, q, r, "signed"), d_out=G.make_grad(43, 1, B, D))
    for mode in ("dense", "sgd", "adagrad"):
        got, again, orc = run_case(c, mode, plan_shared=True), run_case(c, mode, plan_shared=True), oracle_case(c, mode)
        gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
        for k in range(len(p)):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"hot {p} grad{k}")
                assert np.array_equal(got["grads"][k], again["grads"][k]), "not deterministic"
            elif mode == "sgd":
                assert_close(got["cores"][k], orc["cores"][k], f"hot {p} sgd core{k}")
                assert np.array_equal(got["cores"][k], again["cores"][k]), "not deterministic"
            else:
                assert_close(got["state"][k], orc["state"][k], f"hot {p} state{k}")
                assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"hot {p} adagrad core{k}")


@pytest.mark.parametrize("q", [[4, 4, 4], [2, 4, 4]])
def test_benchmark_shape_large_batch_variant(q):
    """the benchmark shape (and its q0 = 2 sibling) at a batch that no longer fits the single-launch plan
    (> 16384 lookups): forward, dense gradients and fused SGD against the oracle"""
    p, r = [9, 8, 7], [1, 32, 32, 1]
    E_, D, B = int(np.prod(p)), int(np.prod(q)), 6800
    idx, off = G.make_bags(51, B, E_, 20, 2, 1)
    assert idx.size > 131072  # ... and the plan's chunks hold four 16- / 32-lookup sub-chunks (kernel variant MULTI)
    c = dict(tables=1, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(52, 1, p, q, r, "signed"), d_out=G.make_grad(53, 1, B, D))
    for mode in ("dense", "sgd"):
        got, orc = run_case(c, mode, plan_shared=True), oracle_case(c, mode)
        assert_close(got["out"], orc["out"], "large-batch variant out")
        for k in range(3):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"large-batch variant grad{k}")
            else:
                assert_close(got["cores"][k], orc["cores"][k], f"large-batch variant sgd core{k}")


@pytest.mark.parametrize("mc,tables", [(128, 1), (64, 3), (256, 1)])
def test_bwd32_experiment_vs_oracle(mc, tables):
    """round 6's experimental backward of the benchmark shape at large batches (csrc/ttx_tt_spec.inc bwd32_kernel: eight lookups per
    wave on v_mfma_f32_32x32x2, both register contractions on v_mfma_f32_4x4x1, persistent work-groups that take their chunks from
    a counter; TEST BUILD only, ttx_debug_bwd32 -- measured slower than spec_bwd_kernel): dense gradients and fused
    SGD against the oracle, bit-identical from run to run and to itself at another chunk length's ... thin cores (their sums do not
    depend on the chunking); slices of 1 .. 3000 lookups: partial sub-chunks, waves without lookups, chunks of one sub-chunk"""
    import tt_embeddings as E

    p, q, r = [9, 8, 7], [4, 4, 4], [1, 32, 32, 1]
    E_, D, B = int(np.prod(p)), int(np.prod(q)), 6800 // tables + 1
    idx, off = G.make_bags(61 + tables, B, E_, 20, 2, tables)
    assert idx.size > 131072
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(62, tables, p, q, r, "signed"), d_out=G.make_grad(63, tables, B, D))
    plain = run_case(c, "dense", plan_shared=True)
    E.debug_bwd32(mc)
    try:
        got, again = run_case(c, "dense", plan_shared=True), run_case(c, "dense", plan_shared=True)
        sgd = run_case(c, "sgd", plan_shared=True)
    finally:
        E.debug_bwd32(0)
    orc, orc_sgd = oracle_case(c, "dense"), oracle_case(c, "sgd")
    for k in range(3):
        assert_close(got["grads"][k], orc["grads"][k], f"bwd32 mc={mc} grad{k}")
        assert np.array_equal(got["grads"][k], again["grads"][k]), "not deterministic"
        assert_close(sgd["cores"][k], orc_sgd["cores"][k], f"bwd32 mc={mc} sgd core{k}")
        assert_close(got["grads"][k], plain["grads"][k], f"bwd32 mc={mc} vs spec_bwd_kernel grad{k}")


@pytest.mark.parametrize("ranks,q", [([32, 32], [4, 4, 4]), ([16, 16], [4, 4, 4]), ([32, 32], [4, 4, 8]), ([16, 16], [4, 4, 8]),
                                      ([64, 64], [4, 4, 8]), ([64, 64], [4, 4, 4]), ([32, 32], [2, 4, 4]), ([16, 16], [2, 4, 4]), ([64, 64], [2, 4, 4]),
                                      ([32, 32], [4, 8, 8]), ([64, 64], [4, 8, 8]), ([32, 32], [2, 2, 4]), ([64, 64], [2, 2, 4]), ([16, 16], [2, 2, 4]),
                                      ([16, 16], [4, 8, 8]), ([128, 128], [4, 4, 4]), ([128, 128], [4, 4, 8]), ([128, 128], [4, 8, 8])])
def test_specialised_shapes_vs_oracle_and_generic(ranks, q):
    """the shape-specialised wave-independent kernels (ttx_tt_spec.inc): against the oracle,
    and against the generic kernels (forced with the debug knob) on the same inputs;
    slices with 1..70 lookups exercise partial groups of 4 and chunks of 32 (16 for r = 64, which
    also walks core 1 in four column passes); q0 = 2 (the reference's default factoring of D = 32): a 16-row
    tile is eight lookups of two rows, a lane's accumulator rows belong to two lookups"""
    import tt_embeddings as E

    p = [6, 5, 7]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    for tables, B, pf, std in ((1, 90, 3, 2), (3, 40, 5, 4), (1, 7, 1, 0)):
        idx, off = G.make_bags(11 + B, B, E_, pf, std, tables)
        c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                 cores=G.make_cores(12 + B, tables, p, q, r, "signed"), d_out=G.make_grad(13, tables, B, D))
        for mode in ("dense", "sgd", "adagrad"):
            got = run_case(c, mode, plan_shared=True)
            orc = oracle_case(c, mode)
            E.debug_skip(256)  # generic kernels (r = 64 with q1 = 8: a 128 KB core_1 slice, walked in blocks)
            try:
                gen = run_case(c, mode, plan_shared=False)
            finally:
                E.debug_skip(0)
            assert_close(got["out"], orc["out"], f"spec {ranks}{q} out vs oracle")
            assert_close(got["out"], gen["out"], f"spec {ranks}{q} out vs generic")
            gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
            for k in range(3):
                if mode == "dense":
                    assert_close(got["grads"][k], orc["grads"][k], f"spec {ranks}{q} grad{k} vs oracle")
                    assert_close(got["grads"][k], gen["grads"][k], f"spec {ranks}{q} grad{k} vs generic")
                elif mode == "sgd":
                    assert_close(got["cores"][k], orc["cores"][k], f"spec {ranks}{q} sgd core{k}")
                else:
                    assert_close(got["state"][k], orc["state"][k], f"spec {ranks}{q} state{k}")
                    assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"spec {ranks}{q} adagrad core{k}")


def _check_modes(c, what, modes=("dense", "sgd", "adagrad")):
    for mode in modes:
        got = run_case(c, mode, plan_shared=(mode != "sgd"))
        orc = oracle_case(c, mode)
        assert_close(got["out"], orc["out"], f"{what} out")
        gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
        for k in range(c["T"]):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"{what} grad{k}")
            elif mode == "sgd":
                assert_close(got["cores"][k], orc["cores"][k], f"{what} sgd core{k}")
            else:
                assert_close(got["state"][k], orc["state"][k], f"{what} state{k}")
                assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"{what} adagrad core{k}")


@pytest.mark.parametrize("q,ranks,budget", [
    ([4, 4, 4], [80, 48], 24), ([3, 5, 2], [70, 13], 24), ([4, 4, 4], [40, 40], 32), ([2, 3, 2, 3], [40, 24, 20], 24),
    ([3, 4, 5, 7], [13, 12, 7], 24), ([2, 2, 3, 2], [33, 9, 35], 16), ([4, 8], [72], 24), ([3, 5], [45], 20), ([4, 4, 8], [64, 32], 40)])
def test_block_walk_under_a_small_lds_budget(q, ranks, budget):
    """The generic kernels walk a core_1 slice in K blocks x column passes when it does not fit the LDS
    (csrc/ttx_tt_generic.inc; the reference contracts any (m, k, n): tt_embeddings_cuda.cu:993-1054).  A small
    test budget sends small shapes through that walk: forward, dense / SGD / Adagrad backward against the oracle,
    T = 2, 3, 4, aligned and odd extents, one and three tables, partial chunks."""
    import tt_embeddings as E

    T = len(q)
    p = [6, 5, 7, 3][:T]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    E.debug_lds_budget(budget * 1024)
    E.debug_skip(256)  # (some of these shapes would take a padded shape-specialised kernel: this is about the generic ones)
    try:
        tiles = E.debug_tiles(1, p, q, r)
        assert tiles["MC"] > 0 and tiles["ncp"] * tiles["nkb"] > 1 and tiles["bytes"] <= budget * 1024, tiles
        for tables, B, pf, std in ((1, 70, 4, 3), (3, 30, 3, 2), (1, 5, 1, 0)):
            idx, off = G.make_bags(21 + B, B, E_, pf, std, tables)
            c = dict(tables=tables, T=T, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                     cores=G.make_cores(22 + B, tables, p, q, r, "signed"), d_out=G.make_grad(23, tables, B, D))
            _check_modes(c, f"walk q={q} r={ranks} {tiles} tables={tables}")
    finally:
        E.debug_lds_budget(0)
        E.debug_skip(0)


@pytest.mark.parametrize("q,ranks", [([4, 4, 4], [128, 128]), ([4, 4, 4], [96, 96]), ([4, 4, 4], [80, 80]), ([2, 8, 8], [64, 64]),
                                      ([4, 4, 4, 4], [128, 128, 128]), ([4, 4, 4], [256, 256]), ([2, 4, 4], [72, 200]),
                                      ([4, 16], [300]), ([3, 4, 5], [130, 67])])
def test_large_rank_shapes(q, ranks):
    """Shapes whose core_1 slice is beyond the LDS (ranks >= 80 at T >= 3; r = 64 with q = [2, 8, 8]): refused with
    TTX_EUNSUPPORTED until round 3, now walked in blocks -- the reference takes any ranks through
    cublasGemmBatchedEx (tt_embeddings_cuda.cu:993-1054, :529-591).  Forward + all three backward modes vs the oracle."""
    import tt_embeddings as E

    T = len(q)
    p = [6, 5, 7, 3][:T]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    E.debug_skip(256)  # (r = 64 with q = [2,8,8] also fits a padded specialised kernel: the generic walk is meant here)
    try:
        tiles = E.debug_tiles(1, p, q, r)
        assert tiles["MC"] > 0 and tiles["bytes"] <= 160 * 1024, tiles
        _large_rank_cases(E, T, p, q, r, ranks, E_, D, tiles)
    finally:
        E.debug_skip(0)


def _large_rank_cases(E, T, p, q, r, ranks, E_, D, tiles):
    for tables, B, pf, std in ((1, 60, 4, 3), (2, 20, 3, 2)):
        idx, off = G.make_bags(31 + B, B, E_, pf, std, tables)
        c = dict(tables=tables, T=T, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                 cores=G.make_cores(32 + B, tables, p, q, r, "signed"), d_out=G.make_grad(33, tables, B, D))
        _check_modes(c, f"large ranks q={q} r={ranks} {tiles} tables={tables}", modes=("dense", "sgd") if tables == 1 else ("adagrad",))


def test_edge_cases():
    import tt_embeddings as E

    p, q, r = [7, 9, 11], [3, 4, 5], [1, 13, 12, 1]
    cores = G.make_cores(5, 1, p, q, r)
    g = O.make_geom(1, p, q, r)
    Lt = torch.zeros(3, dtype=torch.int64, device=dev())
    # nnz == 0 -> zeros (cu:981-985)
    e = torch.empty(0, dtype=torch.int64, device=dev())
    out = E.tt_forward(1000, 1, 4, 60, p, q, r, Lt, 0, e, e, e, [t(x) for x in cores])
    assert out.shape == (1, 4, 60) and float(out.abs().max()) == 0.0
    grads = E.tt_dense_backward(1000, 60, p, q, r, Lt, 0, e, e, e, torch.zeros(1, 4, 60, device=dev()), [t(x) for x in cores])
    assert all(float(gk.abs().max()) == 0.0 for gk in grads)
    # one bag holding every row twice, other bags empty; first/last index of the table
    E_ = 7 * 9 * 11
    idx = np.concatenate([np.arange(E_), np.arange(E_)[::-1], [0, E_ - 1]]).astype(np.int64)
    off = np.array([0, 0, idx.size - 2, idx.size - 2, idx.size], dtype=np.int64)
    c = dict(tables=1, T=3, p=p, q=q, r=r, B=4, D=60, indices=idx, offsets=off, cores=cores, d_out=G.make_grad(6, 1, 4, 60))
    for mode in ("dense", "sgd"):
        got, orc = run_case(c, mode), oracle_case(c, mode)
        assert_close(got["out"], orc["out"], "edge out")
        for k in range(3):
            if mode == "dense":
                assert_close(got["grads"][k], orc["grads"][k], f"edge grad{k}")
            else:
                assert_close(got["cores"][k], orc["cores"][k], f"edge sgd core{k}")


def test_toy_d_not_multiple_of_4(small_cases):
    """BASELINE config 1 (README toy example, D = 3): the reference rejects
    D % 4 != 0 (cu:989); this build supports it."""
    c = small_cases["cfg1_toy"]
    got = run_case(c, "dense")
    assert_close(got["out"], c["out"], "toy out")
    for k in range(3):
        assert_close(got["grads"][k], c["grads"][k], f"toy grad{k}")


def test_errors():
    import tt_embeddings as E

    p, q, r = [7, 9, 11], [3, 4, 5], [1, 13, 12, 1]
    cores = [t(x) for x in G.make_cores(5, 1, p, q, r)]
    Lt = torch.zeros(3, dtype=torch.int64, device=dev())
    i = torch.zeros(4, dtype=torch.int64, device=dev())
    with pytest.raises(RuntimeError):  # D mismatch
        E.tt_forward(1000, 1, 2, 64, p, q, r, Lt, 4, i, i, i, cores)
    with pytest.raises(RuntimeError):  # CPU tensor
        E.tt_forward(1000, 1, 2, 60, p, q, r, Lt, 4, i.cpu(), i, i, [x.cpu() for x in cores])
    with pytest.raises(RuntimeError):  # int32 indices
        E.tt_forward(1000, 1, 2, 60, p, q, r, Lt, 4, i.int(), i, i, cores)


@pytest.mark.parametrize("ranks,q", [([32, 32], [4, 4, 4]), ([16, 16], [4, 4, 4]), ([64, 64], [4, 4, 8]), ([32, 32], [2, 4, 4]),
                                      ([32, 32], [4, 8, 8]), ([16, 16], [2, 2, 4]), ([16, 16], [4, 4, 8]), ([13, 12], [3, 4, 5])])
def test_pooling_fused_into_the_forward_kernel_is_bit_identical(ranks, q):
    """ttx_tt_forward_o: given the bags' offsets the contraction kernel pools the bags itself -- the lookup that completes
    a bag sums its rows in index order, the order of reduce_output_kernel (tt_embeddings_cuda.cu:920-962) -- instead of a
    pooling launch of its own.  Output identical BIT FOR BIT to the unfused path: ragged and empty bags, several tables,
    one long bag next to many short ones (the work-groups finish at very different times), per_sample_weights, and again
    on the same counters (they must be left zeroed).  Shapes without a fused variant take the launch and agree trivially."""
    import tt_embeddings as E

    p = [6, 5, 7]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    Lt = torch.zeros(3, dtype=torch.int64, device=dev())
    e0 = torch.empty(0, dtype=torch.int64, device=dev())
    e1 = torch.empty(0, dtype=torch.int32, device=dev())
    rs = np.random.RandomState(5)
    cases = []
    for tables, B, pf, std in ((1, 90, 3, 2), (3, 40, 5, 4), (1, 7, 1, 0), (2, 300, 20, 0)):
        cases.append((tables, B) + G.make_bags(41 + B, B, E_, pf, std, tables))
    # one bag of 3000 lookups among 200 short ones; half of the bags empty
    lens = np.where(rs.rand(201) < 0.5, 0, rs.randint(1, 6, size=201))
    lens[77] = 3000
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cases.append((1, 201, rs.randint(0, E_, size=int(off[-1])).astype(np.int64), off))
    for tables, B, idx, off in cases:
        cores = [t(x) for x in G.make_cores(42 + B, tables, p, q, r, "signed")]
        ti, to = t(idx), t(off)
        _, rowidx, tableidx, _, _ = E.preprocess_indices_sync(ti, to, tables, True, e0, e1)
        nnz = ti.numel()
        plan = E.make_plan(tables, p, q, r, nnz, ti, tableidx, rowidx)
        ref = E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, ti, rowidx, tableidx, cores, plan=plan)
        for rep in range(3):
            got = E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, ti, rowidx, tableidx, cores, plan=plan if rep else None, offsets=to)
            assert torch.equal(got, ref), f"fused pooling differs (tables={tables} B={B} rep={rep})"
        w = t(rs.rand(nnz).astype(np.float32))
        ref_w = E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, ti, rowidx, tableidx, cores, plan=plan, per_sample_weights=w)
        got_w = E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, ti, rowidx, tableidx, cores, plan=plan, offsets=to, per_sample_weights=w)
        assert torch.equal(got_w, ref_w), "weighted fused pooling differs"
        arr = E._arrive_cache[(0, E._stream(dev()))]
        assert int(arr.abs().sum()) == 0, "arrival counters were not left zeroed"
    c = dict(tables=1, T=3, p=p, q=q, r=r, B=cases[0][1], D=D, indices=cases[0][2], offsets=cases[0][3],
             cores=G.make_cores(42 + cases[0][1], 1, p, q, r, "signed"), d_out=G.make_grad(43, 1, cases[0][1], D))
    got = E.tt_forward(1000, 1, c["B"], D, p, q, r, Lt, c["indices"].size, t(c["indices"]),
                       *E.preprocess_indices_sync(t(c["indices"]), t(c["offsets"]), 1, True, e0, e1)[1:3], [t(x) for x in c["cores"]],
                       offsets=t(c["offsets"]))
    assert_close(got.cpu().numpy(), oracle_case(c, "fwd")["out"], "fused pooling vs oracle")


def test_fused_pooling_stress_across_xcds():
    """Round 4 verdict: the arrival protocol of the fused pooling (sc1 stores, one relaxed agent-scope fetch_add per lookup, the
    completing lookup's sc1 loads) had no stress test with work-groups on different XCDs finishing out of order AT SCALE.  The
    benchmark's geometry (220 pivot slices: a bag's lookups sit in that many different work-groups, dealt round-robin over the 8
    XCDs), 6000 bags of skewed lengths -- most short, a tail of hundreds of lookups, a tenth empty -- 120,000+ lookups, 100
    repeats on the same counters: every repeat bit-identical to the pooling launch, the counters left zeroed every time."""
    import tt_embeddings as E

    p, q, r = [200, 220, 250], [4, 4, 4], [1, 32, 32, 1]
    E_, D, B = int(np.prod(p)), 64, 6000
    rs = np.random.RandomState(77)
    lens = np.minimum(rs.zipf(1.6, size=B) * 3, 900)      # skewed: median 3, a tail of hundreds
    lens[rs.rand(B) < 0.1] = 0
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(off[-1])
    assert nnz > 100_000 and lens.max() >= 300
    # a skewed index stream on top: hot pivot slices are thousands of lookups = hundreds of chunks that finish at different times
    idx = np.where(rs.rand(nnz) < 0.3, rs.zipf(1.3, size=nnz) % E_, rs.randint(0, E_, size=nnz)).astype(np.int64)
    Lt = torch.zeros(3, dtype=torch.int64, device=dev())
    e0 = torch.empty(0, dtype=torch.int64, device=dev())
    e1 = torch.empty(0, dtype=torch.int32, device=dev())
    cores = [t(x) for x in G.make_cores(78, 1, p, q, r, "signed")]
    ti, to = t(idx), t(off)
    _, rowidx, tableidx, _, _ = E.preprocess_indices_sync(ti, to, 1, True, e0, e1)
    plan = E.make_plan(1, p, q, r, nnz, ti, tableidx, rowidx)
    ref = E.tt_forward(1000, 1, B, D, p, q, r, Lt, nnz, ti, rowidx, tableidx, cores, plan=plan)
    bad = 0
    for rep in range(100):
        got = E.tt_forward(1000, 1, B, D, p, q, r, Lt, nnz, ti, rowidx, tableidx, cores, plan=plan, offsets=to)
        if rep == 0:
            arr = E._arrive_cache[(0, E._stream(dev()))]
        bad += int(not torch.equal(got, ref)) + int(arr.abs().sum().item() != 0)
    assert bad == 0, f"{bad} of 100 repeats differ from the pooling launch or left counters behind"


@pytest.mark.parametrize("q,ranks", [([3, 4, 5], [13, 12]), ([4, 4, 4], [13, 12]), ([4, 4, 4], [24, 24]), ([4, 4, 8], [48, 40]),
                                      ([4, 4, 4], [8, 8]), ([1, 3, 4], [16, 16]), ([2, 3, 3], [20, 32]), ([4, 5, 7], [60, 40]),
                                      ([3, 8, 8], [64, 50]), ([2, 2, 4], [12, 12]), ([4, 4, 4], [16, 32]), ([2, 8, 8], [64, 64]),
                                      ([4, 4, 4], [96, 96]), ([4, 4, 4], [80, 100]), ([3, 4, 7], [72, 128]), ([2, 4, 4], [128, 128]), ([4, 6, 8], [100, 90]),
                                      ([4, 8, 16], [32, 32]), ([4, 8, 12], [24, 32]), ([3, 8, 10], [16, 16]), ([4, 8, 16], [16, 16]),
                                      ([4, 8, 16], [64, 64]), ([4, 8, 12], [48, 64]), ([4, 6, 9], [40, 56]),  # (q2 <= 16 at ranks <= 64)
                                      ([4, 9, 10], [32, 32]), ([4, 16, 16], [32, 32]), ([4, 10, 11], [16, 16]), ([3, 12, 12], [24, 32]),
                                      ([4, 16, 16], [16, 16]), ([2, 10, 12], [20, 28]),  # (round 5: q1 up to 16 at ranks <= 32)
                                      ([4, 4, 17], [32, 32]), ([2, 8, 19], [16, 16]), ([4, 8, 23], [32, 32]), ([3, 8, 17], [24, 32]),
                                      ([4, 4, 32], [32, 32]), ([4, 8, 32], [16, 16]), ([4, 7, 29], [32, 20]), ([1, 4, 31], [16, 16]),  # (... q2 up to 32)
                                      ([4, 16, 16], [64, 64]), ([4, 9, 10], [48, 64]),  # (q1 up to 16 at r = 64: gradient rows per pass)
                                      ([1, 16, 23], [32, 32]), ([2, 16, 29], [16, 16]), ([4, 16, 32], [32, 32]), ([1, 16, 31], [20, 24])])  # (q1 = 16 AND q2 up to 32)
def test_padded_shapes_run_on_the_specialised_kernels(q, ranks):
    """Round 3: a T = 3 geometry with q0 <= 4, q1 <= 8, q2 <= 8 and ranks <= 128 that is NOT one of the exact shapes --
    ranks that are not multiples of 16 (the reference tests' 13 / 12, tt_embeddings_test.py:65-70), factorings like [3, 4, 5] --
    runs on the smallest shape-specialised kernel that holds it (PAD variants, csrc/ttx_tt_spec.inc: the global loads read zeros
    outside the real extents, the stores skip them).  Against the oracle and against the generic kernels; forward, dense / SGD /
    Adagrad; one and three tables, ragged bags, partial groups."""
    import tt_embeddings as E

    p = [6, 5, 7]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    assert E.debug_tiles(1, p, q, r)["MC"] == 0, "the geometry is expected to take a (padded) specialised kernel"
    for tables, B, pf, std in ((1, 90, 3, 2), (3, 40, 5, 4), (1, 7, 1, 0)):
        idx, off = G.make_bags(51 + B, B, E_, pf, std, tables)
        c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                 cores=G.make_cores(52 + B, tables, p, q, r, "signed"), d_out=G.make_grad(53, tables, B, D))
        for mode in ("dense", "sgd", "adagrad"):
            got = run_case(c, mode, plan_shared=True)
            orc = oracle_case(c, mode)
            E.debug_skip(256)  # generic kernels
            try:
                gen = run_case(c, mode, plan_shared=False)
            finally:
                E.debug_skip(0)
            assert_close(got["out"], orc["out"], f"padded {ranks}{q} out vs oracle")
            assert_close(got["out"], gen["out"], f"padded {ranks}{q} out vs generic")
            gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
            for k in range(3):
                if mode == "dense":
                    assert_close(got["grads"][k], orc["grads"][k], f"padded {ranks}{q} grad{k} vs oracle")
                    assert_close(got["grads"][k], gen["grads"][k], f"padded {ranks}{q} grad{k} vs generic")
                elif mode == "sgd":
                    assert_close(got["cores"][k], orc["cores"][k], f"padded {ranks}{q} sgd core{k}")
                else:
                    assert_close(got["state"][k], orc["state"][k], f"padded {ranks}{q} state{k}")
                    assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"padded {ranks}{q} adagrad core{k}")


@pytest.mark.parametrize("q,ranks", [([2, 4, 4, 2], [32, 32, 32]), ([4, 4, 2, 2], [16, 32, 16]), ([3, 4, 2, 3], [13, 12, 7]),
                                      ([2, 2, 4, 2], [16, 16, 8]), ([4, 4, 4, 2], [32, 32, 8]), ([2, 8, 2, 4], [64, 40, 20]),
                                      ([2, 3, 1, 5], [20, 30, 33]),
                                      # merged last factor up to 16 (templates with q2 <= 16 at ranks <= 32): the default four-core
                                      # factorings of D = 256 / 128, and odd ones
                                      ([4, 4, 4, 4], [32, 32, 32]), ([2, 4, 4, 4], [16, 16, 16]), ([4, 8, 5, 3], [32, 24, 9]),
                                      ([3, 5, 2, 6], [20, 32, 16]),
                                      # merged last factor up to 32 (round 5: templates with q2 = 32 at ranks <= 32): the default
                                      # four-core factorings of D = 320 / 512 / 768
                                      ([4, 4, 4, 5], [16, 16, 16]), ([4, 4, 4, 8], [32, 32, 32]), ([4, 6, 4, 8], [32, 24, 16])])
def test_four_cores_run_on_the_three_core_kernels(q, ranks):
    """Round 4: a T = 4 geometry with q2 q3 <= 16 (q3 <= 8) runs on the shape-specialised three-core kernels -- the last two cores of a
    lookup are contracted first (per lookup, M = core_2[i2] * core_3[i3]: matrix-chain order, a tenth of the multiply-adds the
    reference's left-to-right order spends on core 2), the three-core kernel reads M where it reads core 2's slice, and the
    backward's d M is turned into the partial rows of cores 2 and 3 (csrc/ttx_tt.hip t4_merge_kernel / t4_unmerge_kernel).
    Against the oracle (left to right, as the reference) and against the generic kernels; forward, dense / SGD / Adagrad; one and
    three tables, ragged bags, partial groups."""
    import tt_embeddings as E

    p = [4, 5, 3, 4]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    assert E.debug_tiles(1, p, q, r)["MC"] == 0, "the geometry is expected to take the three-core kernels"
    for tables, B, pf, std in ((1, 90, 3, 2), (3, 40, 5, 4), (1, 7, 1, 0)):
        idx, off = G.make_bags(61 + B, B, E_, pf, std, tables)
        c = dict(tables=tables, T=4, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                 cores=G.make_cores(62 + B, tables, p, q, r, "signed"), d_out=G.make_grad(63, tables, B, D))
        for mode in ("dense", "sgd", "adagrad"):
            got = run_case(c, mode, plan_shared=True)
            orc = oracle_case(c, mode)
            E.debug_skip(256)  # generic kernels
            try:
                gen = run_case(c, mode, plan_shared=False)
            finally:
                E.debug_skip(0)
            assert_close(got["out"], orc["out"], f"T=4 {ranks}{q} out vs oracle")
            assert_close(got["out"], gen["out"], f"T=4 {ranks}{q} out vs generic")
            gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
            for k in range(4):
                if mode == "dense":
                    assert_close(got["grads"][k], orc["grads"][k], f"T=4 {ranks}{q} grad{k} vs oracle")
                    assert_close(got["grads"][k], gen["grads"][k], f"T=4 {ranks}{q} grad{k} vs generic")
                elif mode == "sgd":
                    assert_close(got["cores"][k], orc["cores"][k], f"T=4 {ranks}{q} sgd core{k}")
                else:
                    assert_close(got["state"][k], orc["state"][k], f"T=4 {ranks}{q} state{k}")
                    assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"T=4 {ranks}{q} adagrad core{k}")


@pytest.mark.parametrize("q,ranks", [([8, 8], [32]), ([4, 16], [128]), ([2, 2], [4]), ([16, 8], [64]), ([5, 7], [20]), ([3, 4], [12]),
                                      # r1 % 4 != 0 (zero-padded k tiles, scalar loads / stores of core 0's rows): the reference tests' r = 13
                                      ([3, 4], [13]), ([5, 7], [10]), ([16, 16], [17]), ([2, 3], [1]),
                                      # q up to 32 (round 5): the default two-core factorings of D = 512 / 640 / 1024
                                      ([16, 32], [32]), ([20, 32], [24]), ([32, 32], [64])])
def test_two_cores_on_the_dedicated_kernels(q, ranks):
    """Round 4: a T = 2 geometry with r1 <= 128, q <= 16 runs on the dedicated two-core kernels (csrc/ttx_tt.hip
    t2_fwd_kernel / t2_bwd_kernel: a lookup is one [q0 x r1] x [r1 x q1] product -- byte work, no matrix tiles to pad).  Against the
    oracle and against the generic kernels; forward, dense / SGD / Adagrad; one and three tables, ragged bags, partial chunks."""
    import tt_embeddings as E

    p = [9, 8]
    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    assert E.debug_tiles(1, p, q, r)["MC"] == 0, "the geometry is expected to take the two-core kernels"
    for tables, B, pf, std in ((1, 90, 3, 2), (3, 40, 5, 4), (1, 7, 1, 0)):
        idx, off = G.make_bags(71 + B, B, E_, pf, std, tables)
        c = dict(tables=tables, T=2, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                 cores=G.make_cores(72 + B, tables, p, q, r, "signed"), d_out=G.make_grad(73, tables, B, D))
        for mode in ("dense", "sgd", "adagrad"):
            got = run_case(c, mode, plan_shared=True)
            orc = oracle_case(c, mode)
            E.debug_skip(256)  # generic kernels
            try:
                gen = run_case(c, mode, plan_shared=False)
            finally:
                E.debug_skip(0)
            assert_close(got["out"], orc["out"], f"T=2 {ranks}{q} out vs oracle")
            assert_close(got["out"], gen["out"], f"T=2 {ranks}{q} out vs generic")
            gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
            for k in range(2):
                if mode == "dense":
                    assert_close(got["grads"][k], orc["grads"][k], f"T=2 {ranks}{q} grad{k} vs oracle")
                    assert_close(got["grads"][k], gen["grads"][k], f"T=2 {ranks}{q} grad{k} vs generic")
                elif mode == "sgd":
                    assert_close(got["cores"][k], orc["cores"][k], f"T=2 {ranks}{q} sgd core{k}")
                else:
                    assert_close(got["state"][k], orc["state"][k], f"T=2 {ranks}{q} state{k}")
                    assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], f"T=2 {ranks}{q} adagrad core{k}")


@pytest.mark.parametrize("T,p,q,ranks", [(2, [600, 700], [8, 8], [32]), (2, [1100, 90], [4, 16], [16]), (3, [400, 500, 450], [4, 4, 4], [4, 4])])
def test_reduce_apply_with_a_wave_per_small_slice(T, p, q, ranks):
    """Round 6: when every slice is at most 64 float4 lanes and there are many of them (two cores over millions of rows; three at tiny
    ranks), reduce_apply hands FOUR slices to a work-group, one per wave -- a slice's partial rows summed by its wave in index
    order, no work-group barrier, no LDS fold -- and numbers the hot slices' segment work-groups behind the packed ones.  Against
    the oracle for the dense gradient, SGD and Adagrad, on a uniform stream and on a skewed one (hot slices: segment work-groups in
    the same launch), and against the launch with a work-group per slice (ttx_debug_skip bit 16): equal up to the order of
    addition; two runs of the packed launch are bit-identical."""
    import tt_embeddings as E

    r = [1] + ranks + [1]
    E_, D = int(np.prod(p)), int(np.prod(q))
    for alpha, B, pf in ((1.0, 300, 20), (1.3, 400, 24)):
        idx, off = G.make_bags(91 + B, B, E_, pf, 3, 1)
        if alpha > 1.0:  # a skewed stream: a few rows take most of the lookups -> hot slices in every core
            rs = np.random.RandomState(5)
            idx = (rs.zipf(alpha, size=idx.size).astype(np.int64) * 7919) % E_
        c = dict(tables=1, T=T, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                 cores=G.make_cores(92 + B, 1, p, q, r, "signed"), d_out=G.make_grad(93, 1, B, D))
        for mode in ("dense", "sgd", "adagrad"):
            got = run_case(c, mode, plan_shared=True)
            again = run_case(c, mode, plan_shared=True)
            orc = oracle_case(c, mode)
            E.debug_skip(1 << 16)  # a work-group per slice, as before round 6
            try:
                old = run_case(c, mode, plan_shared=True)
            finally:
                E.debug_skip(0)
            gref = oracle_case(c, "dense")["grads"] if mode == "adagrad" else None
            for k in range(T):
                what = f"T={T} {p}{q}{ranks} alpha={alpha} {mode} core{k}"
                if mode == "dense":
                    assert np.array_equal(got["grads"][k], again["grads"][k]), what + ": run to run"
                    assert_close(got["grads"][k], orc["grads"][k], what + " vs oracle")
                    assert_close(got["grads"][k], old["grads"][k], what + " vs a work-group per slice")
                elif mode == "sgd":
                    assert np.array_equal(got["cores"][k], again["cores"][k]), what + ": run to run"
                    assert_close(got["cores"][k], orc["cores"][k], what + " vs oracle")
                    assert_close(got["cores"][k], old["cores"][k], what + " vs a work-group per slice")
                else:
                    assert_close(got["state"][k], orc["state"][k], what + " state")
                    assert_adagrad_close(got["cores"][k], orc["cores"][k], gref[k], what)
