"""GPU tests of the module surface (TTEmbeddingBag / TableBatchedTTEmbeddingBag)
on the HIP path: the reference's six property tests restated with fixed seeds,
the benchmark configs at full size against the golden vectors, the cache
life-cycle, determinism."""
import os

import numpy as np
import pytest
import torch

import gen_inputs as G
import oracle_lib as O
from test_oracle_golden import big_case, check_big
from util import LR, EPS, adagrad_expected, assert_adagrad_close, assert_close, sgd_expected

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["native", "python"])
def node(request, monkeypatch):
    """every module test runs twice: through the C++ autograd node (csrc/ttx_torch.cpp) and through
    the reference-shaped Python TTLookupFunction (ctypes) -- both end in the same C ABI"""
    import tt_embeddings_ops as ops

    if request.param == "python":
        monkeypatch.setenv("TTX_NO_NATIVE_NODE", "1")
        assert ops._native_node() is None
    else:
        monkeypatch.delenv("TTX_NO_NATIVE_NODE", raising=False)
        assert ops._native_node() is not None, "ttx_torch.so not built / not importable on this box"
    return request.param


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def module_for(c, **kw):
    import tt_embeddings_ops as ops

    m = ops.TableBatchedTTEmbeddingBag(c["tables"], int(np.prod(c["p"])), c["D"], c["r"][1:-1], c["p"], c["q"],
                                       weight_dist="uniform", use_cache=False, device=DEV, **kw)
    with torch.no_grad():
        for dst, src in zip(m.tt_cores, c["cores"]):
            dst.copy_(t(src))
    return m


def test_module_matches_round4_golden(round4_cases):
    """the module on the round-4 golden vectors: q0 = 5 / 7 through the zero-padded copy of core 0, q0 = 8 as part lookups, the
    rest on whatever the engine picks -- forward, dense gradients, fused SGD against what the reference's Python gives"""
    import tt_embeddings_ops as ops

    for name, c in round4_cases.items():
        m = module_for(c, sparse=False)
        if ops._native_node() is not None:
            want = {"t3_q5": (8, 2), "t3_q7": (8, 2), "t3_q8": (0, 2)}.get(name[:5], (0, 0))
            assert (m._pad0, m._split0) == want, f"{name}: {(m._pad0, m._split0)}"
        out = m(t(c["indices"]), t(c["offsets"]))
        assert_close(out.detach().cpu().numpy(), c["out"], f"{name} out")
        out.backward(t(c["d_out"]))
        for k in range(c["T"]):
            assert_close(m.tt_cores[k].grad.cpu().numpy(), c["grads"][k], f"{name} grad{k}")
        m = module_for(c, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)
        m(t(c["indices"]), t(c["offsets"])).backward(t(c["d_out"]))
        for k, e in enumerate(sgd_expected(c["cores"], c["grads"])):
            assert_close(m.tt_cores[k].detach().cpu().numpy(), e, f"{name} sgd{k}")


def test_module_matches_golden(small_cases):
    import tt_embeddings_ops as ops

    for name, c in small_cases.items():
        m = module_for(c, sparse=False)
        out = m(t(c["indices"]), t(c["offsets"]))
        assert_close(out.detach().cpu().numpy(), c["out"], f"{name} out")
        out.backward(t(c["d_out"]))
        for k in range(c["T"]):
            assert_close(m.tt_cores[k].grad.cpu().numpy(), c["grads"][k], f"{name} grad{k}")
        m = module_for(c, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)
        m(t(c["indices"]), t(c["offsets"])).backward(t(c["d_out"]))
        for k, e in enumerate(sgd_expected(c["cores"], c["grads"])):
            assert_close(m.tt_cores[k].detach().cpu().numpy(), e, f"{name} sgd{k}")
        m = module_for(c, sparse=True, optimizer=ops.OptimType.EXACT_ADAGRAD, learning_rate=LR, eps=EPS)
        m(t(c["indices"]), t(c["offsets"])).backward(t(c["d_out"]))
        exp, st = adagrad_expected(c["cores"], c["grads"])
        for k in range(c["T"]):
            assert_close(m.optimizer_state[k].cpu().numpy(), st[k], f"{name} state{k}")
            assert_adagrad_close(m.tt_cores[k].detach().cpu().numpy(), exp[k], c["grads"][k], f"{name} ada{k}")


def test_vs_nn_embedding_bag_on_full_weight():
    """tt_embeddings_test.py:62-107 literally: compare with nn.EmbeddingBag(_weight=full_weight())"""
    import tt_embeddings_ops as ops

    for T in (2, 3, 4):
        p, q, r = G.test_shape(T)
        E_, D = int(np.prod(p)), int(np.prod(q))
        torch.manual_seed(T)
        m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=False, use_cache=False, weight_dist="uniform", device=DEV)
        emb = torch.nn.EmbeddingBag(E_, D, sparse=True, mode="sum", _weight=m.full_weight().detach(), include_last_offset=True).to(DEV)
        idx, off = G.make_bags(T, 300, E_, 5, 4)
        out, ref = m(t(idx), t(off)), emb(t(idx), t(off))
        assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), f"T={T} vs nn.EmbeddingBag")


def _run_big(tag, mode):
    import tt_embeddings_ops as ops

    c, _ = big_case(tag)
    kw = dict(sparse=False) if mode == "dense" else (
        dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR) if mode == "sgd"
        else dict(sparse=True, optimizer=ops.OptimType.EXACT_ADAGRAD, learning_rate=LR, eps=EPS))
    m = module_for(c, **kw)
    out = m(t(c["indices"]), t(c["offsets"]))
    out.backward(t(c["d_out"]))
    res = dict(out=out.detach().cpu().numpy(), cores=[x.detach().cpu().numpy() for x in m.tt_cores])
    if mode == "dense":
        res["grads"] = [x.grad.cpu().numpy() for x in m.tt_cores]
    return res


@pytest.mark.parametrize("tag", ["cfg2", "cfg4", "r128"])
def test_benchmark_configs_full_size(tag):
    """BASELINE configs[1] (cfg2, SGD) and configs[3] (cfg4: ranks 64, D=128,
    Adagrad) at full size against the golden sub-samples / per-slice sums; and ranks [128,128] (the r = 128
    shape-specialised kernels) against what the reference's Python gives for it (SGD and Adagrad)"""
    check_big(tag, _run_big(tag, "dense"), _run_big(tag, "sgd") if tag != "cfg4" else None,
              _run_big(tag, "adagrad") if tag != "cfg2" else None)


def test_full_size_linearity_and_determinism():
    """size-independent properties at cfg2: d(cores) is linear in d_output, and two
    runs give bit-identical results (no atomics on the TT path)"""
    c, _ = big_case("cfg2")
    m = module_for(c, sparse=False)

    def grads(scale):
        for x in m.tt_cores:
            x.grad = None
        m(t(c["indices"]), t(c["offsets"])).backward(t(c["d_out"]) * scale)
        return [x.grad.clone() for x in m.tt_cores]

    g1, g1b, g2 = grads(1.0), grads(1.0), grads(2.0)
    for a, b, d in zip(g1, g1b, g2):
        assert torch.equal(a, b), "backward is not run-to-run deterministic"
        assert torch.equal(a * 2.0, d), "gradient is not exactly linear in d_output (x2 is exact in fp32)"


def test_cache_life_cycle_gpu():
    """warm-up -> populate -> steady state on the GPU; Zipf-skewed lookups
    (BASELINE configs[2] at reduced table size); outputs equal the TT-only module"""
    import tt_embeddings_ops as ops

    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B, Lp = 11000, 64, 128, 10
    rs = np.random.RandomState(0)
    kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform", device=DEV)
    torch.manual_seed(0)
    m = ops.TTEmbeddingBag(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=1e-3, use_cache=True, cache_size=256,
                           hashtbl_size=1 << 16, **kw)
    base = ops.TTEmbeddingBag(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=1e-3, use_cache=False, **kw)
    with torch.no_grad():
        for a, b in zip(base.tt_cores, m.tt_cores):
            a.copy_(b)
    off = t(np.arange(0, B * Lp + 1, Lp, dtype=np.int64))
    batch = lambda: t((rs.zipf(1.2, size=B * Lp).astype(np.int64)) % E_)  # noqa: E731
    grad = t((rs.rand(B, D) * 0.1).astype(np.float32))
    for _ in range(6):
        idx = batch()
        o1, o2 = m(idx, off), base(idx, off)
        assert_close(o1.detach().cpu().numpy(), o2.detach().cpu().numpy(), "warm-up output")
        o1.backward(grad)
        o2.backward(grad)
    # lookups counted (a few may be dropped after 3 probes, like the reference)
    assert 0.99 * 6 * B * Lp <= int(m.cache_freq.sum()) <= 6 * B * Lp
    m.cache_populate()
    assert not m.warmup and int((m.cache_state >= 0).sum()) == min(256, int((m.hashtbl >= 0).sum()))
    idx = batch()
    _, _, _, n_tt, loc = __import__("tt_embeddings").preprocess_indices_sync(idx, off, 1, False, m.hashtbl, m.cache_state)
    assert 0 < n_tt < idx.numel(), "expected a mix of cache hits and TT lookups"
    with torch.no_grad():
        assert_close(m(idx, off).cpu().numpy(), base(idx, off).cpu().numpy(), "steady-state output")
    w0 = m.cache_weight.detach().clone()
    m(idx, off).backward(grad)
    assert not torch.equal(w0, m.cache_weight.detach()), "fused SGD must update the hit cache rows"


def test_inference_no_grad_and_empty_batch():
    import tt_embeddings_ops as ops

    p, q, r = G.test_shape(3)
    m = ops.TTEmbeddingBag(int(np.prod(p)), 60, r, p, q, sparse=False, use_cache=False, weight_dist="uniform", device=DEV)
    with torch.no_grad():
        out = m(torch.empty(0, dtype=torch.int64, device=DEV), torch.zeros(9, dtype=torch.int64, device=DEV))
    assert out.shape == (8, 60) and float(out.abs().max()) == 0.0
    out = m(torch.empty(0, dtype=torch.int32, device=DEV), torch.zeros(9, dtype=torch.int32, device=DEV))
    out.sum().backward()
    assert all(float(c.grad.abs().max()) == 0.0 for c in m.tt_cores)


@pytest.mark.parametrize("p,q,r", [([6, 5, 7], [4, 4, 4], [16, 16]), ([7, 9, 11], [3, 4, 5], [13, 12]), ([5, 8], [3, 4], [6])])
def test_per_sample_weights_vs_nn_embedding_bag(node, p, q, r):
    """nn.EmbeddingBag(mode='sum', per_sample_weights=...) semantics -- forward and the
    cores' gradients against torch's own embedding_bag + autograd on the expanded table; fused SGD = one
    step along that gradient.  (Specialised and generic kernels, T = 2 and 3.)"""
    import tt_embeddings_ops as ops

    E_, D, B = int(np.prod(p)), int(np.prod(q)), 37
    rs = np.random.RandomState(3)
    idx, off = G.make_bags(5, B, E_, 6, 3, 1)
    psw = rs.rand(idx.size).astype(np.float32) * 2 - 0.5
    cores = G.make_cores(8, 1, p, q, r, "signed")
    d_out = G.make_grad(9, 1, B, D)[0]
    kw = dict(use_cache=False, weight_dist="uniform", device=DEV)

    def fresh(**extra):
        m = ops.TTEmbeddingBag(E_, D, r, p, q, **kw, **extra)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, cores):
                dst.copy_(t(src))
        return m

    m = fresh(sparse=False)
    if node == "python":
        with pytest.raises(NotImplementedError):
            m(t(idx), t(off), per_sample_weights=t(psw))
        return
    out = m(t(idx), t(off), per_sample_weights=t(psw))
    # torch reference on the expanded table (differentiable through tt_matrix_to_full)
    ref_cores = [t(c).clone().requires_grad_(True) for c in cores]
    full = ops.tt_matrix_to_full(p, q, [1] + r + [1], ref_cores, [1, 0, 2, 3])
    ref = torch.nn.functional.embedding_bag(t(idx), full, t(off), mode="sum", per_sample_weights=t(psw),
                                            include_last_offset=True)
    assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), "weighted forward")
    out.backward(t(d_out))
    ref.backward(t(d_out))
    for k in range(len(p)):
        assert_close(m.tt_cores[k].grad.cpu().numpy(), ref_cores[k].grad.cpu().numpy(), f"weighted grad{k}")
    ms = fresh(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)
    ms(t(idx), t(off), per_sample_weights=t(psw)).backward(t(d_out))
    for k in range(len(p)):
        assert_close(ms.tt_cores[k].detach().cpu().numpy(), cores[k] - LR * ref_cores[k].grad.cpu().numpy(), f"weighted sgd{k}")
    # the weights' own gradient (d_psw[n] = <d_out[bag(n)], row_n>), dense and fused-optimizer modes
    w_ref = t(psw).clone().requires_grad_(True)
    full2 = ops.tt_matrix_to_full(p, q, [1] + r + [1], [t(c) for c in cores], [1, 0, 2, 3])
    torch.nn.functional.embedding_bag(t(idx), full2, t(off), mode="sum", per_sample_weights=w_ref,
                                      include_last_offset=True).backward(t(d_out))
    for extra in (dict(sparse=False), dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)):
        mw = fresh(**extra)
        w = t(psw).clone().requires_grad_(True)
        mw(t(idx), t(off), per_sample_weights=w).backward(t(d_out))
        assert w.grad is not None, "per_sample_weights that require a gradient must get one"
        assert_close(w.grad.cpu().numpy(), w_ref.grad.cpu().numpy(), f"gradient of per_sample_weights ({extra})")


@pytest.mark.parametrize("optim", ["sgd", "adagrad"])
def test_several_training_steps_track_the_oracle(node, optim):
    """six fused-optimizer steps on changing batches (cache counting, not live): the cores -- and the
    Adagrad state -- after every step against the oracle run on the same stream of batches"""
    import tt_embeddings_ops as ops

    p, q, r = [6, 5, 7], [4, 4, 4], [16, 16]
    E_, D, B = int(np.prod(p)), 64, 64
    cores = G.make_cores(21, 1, p, q, r, "signed")
    opt = ops.OptimType.SGD if optim == "sgd" else ops.OptimType.EXACT_ADAGRAD
    m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=opt, learning_rate=0.05, eps=1e-3, use_cache=True,
                           cache_size=32, hashtbl_size=4096, weight_dist="uniform", device=DEV)
    with torch.no_grad():
        for dst, src in zip(m.tt_cores, cores):
            dst.copy_(t(src))
    g = O.make_geom(1, p, q, r)
    ref = [c.copy() for c in cores]
    state = [np.zeros_like(c) for c in cores]
    for step in range(6):
        idx, off = G.make_bags(100 + step, B, E_, 7, 3, 1)
        d_out = G.make_grad(200 + step, 1, B, D)
        out = m(t(idx), t(off))
        rowidx, tableidx = O.rowidx_from_offsets(off, 1)
        assert_close(out.detach().cpu().numpy(), O.tt_forward(g, B, D, idx, rowidx, tableidx, ref)[0], f"step {step} forward")
        out.backward(t(d_out[0]))
        if optim == "sgd":
            O.tt_backward(g, O.OPTIM_SGD, B, D, 0.05, 0.0, idx, rowidx, tableidx, d_out, ref)
        else:
            O.tt_backward(g, O.OPTIM_ADAGRAD, B, D, 0.05, 1e-3, idx, rowidx, tableidx, d_out, ref, state)
        for k in range(3):
            # (error compounds over the steps: 4x the single-step tolerance)
            a, b = m.tt_cores[k].detach().cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
            tol = 4 * (2e-6 * np.abs(b).max() + 1e-5 * np.abs(b)) * (8 if optim == "adagrad" else 1)
            assert (np.abs(a - b) <= tol).all(), f"step {step} core {k}: max err {np.abs(a - b).max():.3e}"
    assert int(m.cache_freq.sum()) > 0  # the frequency table counted along


@pytest.mark.parametrize("p,q,ranks", [([4, 5, 3, 4], [4, 4, 4, 4], [32, 32, 32]),   # four cores, the MFMA gradient helper
                                        ([4, 5, 3, 4], [3, 4, 2, 3], [13, 12, 7]),    # four cores, the VALU helper
                                        ([9, 8], [16, 32], [32])])                    # two cores
@pytest.mark.parametrize("optim", ["sgd", "adagrad"])
def test_two_and_four_cores_over_several_steps_track_the_oracle(node, p, q, ranks, optim):
    """Round 5: the four-core route keeps state IN THE PLAN between the forward and the backward of a step (the merged last
    cores, csrc/ttx_tt.hip t4_valid) -- so the multi-step life cycle is checked against the oracle run step by step: six
    fused-optimizer steps on changing batches, the last three of them planned ahead in one launch (prefetch_many: three
    plans alive at once while the optimizer rewrites cores 2 / 3 under them).  Two cores ride along (no such state)."""
    import tt_embeddings_ops as ops

    T = len(p)
    E_, D, B = int(np.prod(p)), int(np.prod(q)), 48
    cores = G.make_cores(31, 1, p, q, [1] + ranks + [1], "signed")
    opt = ops.OptimType.SGD if optim == "sgd" else ops.OptimType.EXACT_ADAGRAD
    m = ops.TTEmbeddingBag(E_, D, ranks, p, q, sparse=True, optimizer=opt, learning_rate=0.05, eps=1e-3, use_cache=False,
                           weight_dist="uniform", device=DEV)
    with torch.no_grad():
        for dst, src in zip(m.tt_cores, cores):
            dst.copy_(t(src))
    g = O.make_geom(1, p, q, ranks)
    ref = [c.copy() for c in cores]
    state = [np.zeros_like(c) for c in cores]
    batches = []
    for step in range(6):
        idx, off = G.make_bags(300 + step, B, E_, 5, 3 if step < 3 else 0, 1)  # (a planned round: batches of one size)
        batches.append((t(idx), t(off), idx, off, G.make_grad(400 + step, 1, B, D)))
    for step, (ti, to, idx, off, d_out) in enumerate(batches):
        if step == 3:
            assert m.prefetch_many([(b[0], b[1]) for b in batches[3:]]) is (node == "native")  # (planning ahead needs the C++ node)
        out = m(ti, to)
        rowidx, tableidx = O.rowidx_from_offsets(off, 1)
        assert_close(out.detach().cpu().numpy(), O.tt_forward(g, B, D, idx, rowidx, tableidx, ref)[0], f"step {step} forward")
        out.backward(t(d_out[0]))
        if optim == "sgd":
            O.tt_backward(g, O.OPTIM_SGD, B, D, 0.05, 0.0, idx, rowidx, tableidx, d_out, ref)
        else:
            O.tt_backward(g, O.OPTIM_ADAGRAD, B, D, 0.05, 1e-3, idx, rowidx, tableidx, d_out, ref, state)
        for k in range(T):
            a, b = m.tt_cores[k].detach().cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
            # (Adagrad: lr g / (sqrt(s) + eps) turns a rounding difference of a SMALL g into lr / eps = 50 times as much of the
            # weight -- 16x here, sums over D = 256 at ranks 32; a stale product would be off by the size of an update, ~1e-2)
            tol = 4 * (2e-6 * np.abs(b).max() + 1e-5 * np.abs(b)) * (16 if optim == "adagrad" else 1)
            err = np.abs(a - b)
            assert (err <= tol).all(), f"step {step} core {k}: max err {err.max():.3e} (worst err / tol {(err / tol).max():.2f})"


@pytest.mark.parametrize("p,q,ranks", [([4, 5, 3, 4], [4, 4, 4, 4], [32, 32, 32]), ([4, 5, 3, 4], [3, 4, 2, 3], [13, 12, 7])])
def test_four_cores_two_forwards_outstanding(node, p, q, ranks):
    """Round 6 (advisor): forward A, forward B, backward A, backward B on ONE fused-optimizer module.  Backward A rewrites cores 2 / 3
    while plan B still holds the product of the OLD cores; backward B must recompute it from the current cores, as the reference
    recomputes every intermediate in every backward (tt_embeddings_cuda.cu:419-652) -- csrc/ttx_tt.hip g_t4_epoch.  Checked against
    the oracle run in the same order: backward B with B's output gradient on the cores backward A left.  Both autograd orders:
    `loss.backward()` of a sum (the engine runs B's node first or second) and explicit out.backward() calls."""
    import tt_embeddings_ops as ops

    E_, D, B = int(np.prod(p)), int(np.prod(q)), 40
    cores = G.make_cores(41, 1, p, q, [1] + ranks + [1], "signed")
    g = O.make_geom(1, p, q, ranks)
    (ia, oa), (ib, ob) = G.make_bags(700, B, E_, 5, 2, 1), G.make_bags(701, B, E_, 5, 2, 1)
    da, db = G.make_grad(800, 1, B, D), G.make_grad(801, 1, B, D)
    for order in ("a_then_b", "b_then_a"):
        m = ops.TTEmbeddingBag(E_, D, ranks, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False,
                               weight_dist="uniform", device=DEV)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, cores):
                dst.copy_(t(src))
        out_a, out_b = m(t(ia), t(oa)), m(t(ib), t(ob))
        first, second = ((out_a, da, ia, oa), (out_b, db, ib, ob)) if order == "a_then_b" else ((out_b, db, ib, ob), (out_a, da, ia, oa))
        ref = [c.copy() for c in cores]
        for out, d_out, idx, off in (first, second):
            out.backward(t(d_out[0]))
            rowidx, tableidx = O.rowidx_from_offsets(off, 1)
            O.tt_backward(g, O.OPTIM_SGD, B, D, 0.05, 0.0, idx, rowidx, tableidx, d_out, ref)
        torch.cuda.synchronize()
        for k in range(4):
            a, b = m.tt_cores[k].detach().cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
            tol = 4 * (2e-6 * np.abs(b).max() + 1e-5 * np.abs(b))
            err = np.abs(a - b)
            # (a stale product is off by the size of one update, ~1e-2 of the weights)
            assert (err <= tol).all(), f"{order} core {k}: max err {err.max():.3e} (worst err / tol {(err / tol).max():.2f})"


def test_four_cores_captured_step_tracks_eager(node):
    """... and behind a captured graph (ttx_graph.GraphedStep): plan build, merge, forward, backward and the mark's reset all
    replay from one graph -- ten fused-SGD steps leave the four cores bit-identical to the same steps run eagerly."""
    import tt_embeddings_ops as ops
    import ttx_graph

    p, q, ranks = [4, 5, 3, 4], [4, 4, 4, 4], [32, 32, 32]
    E_, D, B, Lp = int(np.prod(p)), 256, 48, 5

    def fresh():
        m = ops.TTEmbeddingBag(E_, D, ranks, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False,
                               weight_dist="uniform", device=DEV)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, G.make_cores(93, 1, p, q, [1] + ranks + [1], "signed")):
                dst.copy_(t(src))
        return m

    me, mg = fresh(), fresh()
    rs = np.random.RandomState(94)
    off = torch.arange(0, B * Lp + 1, Lp, device=DEV)
    batches = [(t(rs.randint(0, E_, size=B * Lp).astype(np.int64)), off, t((rs.rand(B, D) * 0.1).astype(np.float32))) for _ in range(10)]
    before = [c.detach().clone() for c in mg.tt_cores]
    step = ttx_graph.GraphedStep(lambda i, o, g_: mg(i, o).backward(g_), batches[0], warmup=2)
    with torch.no_grad():
        for c, b in zip(mg.tt_cores, before):
            c.copy_(b)
    for i, o, g_ in batches:
        step(i, o, g_)
        me(i, o).backward(g_)
    torch.cuda.synchronize()
    for a, b in zip(me.tt_cores, mg.tt_cores):
        assert torch.equal(a, b)


def test_drop_in_for_nn_embedding_bag_in_a_dlrm_shaped_model(node):
    """examples/mini_dlrm.py: the TT bags in DLRM's call form (offsets = bag starts only) give the same logits as
    nn.EmbeddingBag tables holding the expanded TT weights, and a few training steps (dense side: torch SGD,
    TT cores: fused SGD in backward) bring the loss down"""
    import importlib.util
    import tt_embeddings_ops as ops

    spec = importlib.util.spec_from_file_location("mini_dlrm", os.path.join(os.path.dirname(__file__), "..", "examples", "mini_dlrm.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    torch.manual_seed(0)
    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_ = 20 * 22 * 25
    tts = [ops.TTEmbeddingBag(E_, 64, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05,
                              use_cache=False, weight_dist="approx-normal", include_last_offset=False, device=DEV)
           for _ in range(3)]
    model = M.MiniDLRM(tts).to(DEV)
    ref = M.MiniDLRM([torch.nn.EmbeddingBag.from_pretrained(e.full_weight().detach().clone(), mode="sum") for e in tts]).to(DEV)
    ref.bot.load_state_dict(model.bot.state_dict())
    ref.top.load_state_dict(model.top.state_dict())

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        dense = torch.rand(64, 13, generator=g).to(DEV)
        sparse = []
        for _ in range(3):
            lengths = torch.randint(1, 6, (64,), generator=g)
            offsets = torch.cat([torch.zeros(1, dtype=torch.int64), lengths.cumsum(0)[:-1]])
            sparse.append((torch.randint(0, E_, (int(lengths.sum()),), generator=g).to(DEV), offsets.to(DEV)))
        return dense, sparse, (dense.sum(1) > 6.5).float()

    dense, sparse, label = batch(0)
    with torch.no_grad():
        assert_close(model(dense, sparse).cpu().numpy(), ref(dense, sparse).cpu().numpy(), "logits vs nn.EmbeddingBag model")
    opt = torch.optim.SGD([p_ for n, p_ in model.named_parameters() if not n.startswith("emb.")], lr=0.1)
    losses = []
    for step in range(30):
        dense, sparse, label = batch(step % 3)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(model(dense, sparse), label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < np.mean(losses[:3]), losses


@pytest.mark.parametrize("optimizer", ["sgd", "adagrad", "dense"])
@pytest.mark.parametrize("shape", ["generic", "spec", "many-slices"])
def test_var_table_batched_lookup(node, optimizer, shape):
    """include/ttx.h ttx_geom::p_tables through ttx_mixed.VarTableTTEmbeddingBag: five tables of three different
    row-factor sets in ONE batched lookup (one plan, one forward, one backward) against a TTEmbeddingBag per
    table with the same cores -- outputs, and cores / state / gradients after a step.  `spec`: the
    specialised r=16 q=[4,4,4] kernels; `generic`: the any-shape ones (wide-digit plan); `many-slices`: the
    multi-pass plan."""
    import tt_embeddings_ops as ops
    import ttx_mixed

    if shape == "spec":
        D, q, r, B = 64, [4, 4, 4], [16, 16], 96
        Es, ps = [9000, 60000, 8000, 900000, 50000], [[20, 22, 25], [40, 40, 40], [20, 22, 25], [100, 100, 100], [40, 40, 40]]
    elif shape == "many-slices":  # 2700 slice ids in core 0: the multi-pass plan (no wide digit)
        D, q, r, B = 12, [2, 3, 2], [4, 5], 300
        Es, ps = [800000, 700000, 90], [[1500, 30, 20], [1200, 25, 30], [4, 5, 5]]
    else:
        D, q, r, B = 12, [2, 3, 2], [4, 5], 50
        Es, ps = [100, 700, 90, 5000, 650], [[4, 5, 5], [8, 9, 10], [4, 5, 5], [20, 16, 16], [8, 9, 10]]
    opt = {"sgd": ops.OptimType.SGD, "adagrad": ops.OptimType.EXACT_ADAGRAD, "dense": ops.OptimType.SGD}[optimizer]
    kw = dict(sparse=optimizer != "dense", optimizer=opt, learning_rate=0.05, eps=1e-4, weight_dist="uniform", device=DEV)
    vm = ttx_mixed.VarTableTTEmbeddingBag(Es, D, r, ps, q, **kw)
    ones = []
    for k in range(len(Es)):
        one = ops.TTEmbeddingBag(Es[k], D, r, ps[k], q, use_cache=False, **kw)
        with torch.no_grad():
            for c in range(3):
                one.tt_cores[c][0].copy_(vm.table_rows(c)[k])
        ones.append(one)
    rs = np.random.RandomState(5)
    grads = t((rs.rand(len(Es), B, D) * 0.1).astype(np.float32))
    for step in range(2):
        idx, off = [], []
        for e in Es:
            lens = rs.randint(0, 9, size=B)
            off.append(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
            idx.append(rs.randint(0, e, size=int(lens.sum())).astype(np.int64))
        mi, mo = ttx_mixed.merge_bags([t(i) for i in idx], [t(o) for o in off], True)
        out = vm(mi, mo)
        assert out.shape == (len(Es), B, D)
        out.backward(grads)
        for k, one in enumerate(ones):
            ref = one(t(idx[k]), t(off[k]))
            assert_close(out[k].detach().cpu().numpy(), ref.detach().cpu().numpy(), f"step {step} table {k} forward")
            ref.backward(grads[k])
        if optimizer == "dense":
            for c in range(3):
                gv = torch.split(vm.tt_cores[c].grad[0], [p[c] for p in ps], dim=0)
                for k, one in enumerate(ones):
                    assert_close(gv[k].cpu().numpy(), one.tt_cores[c].grad[0].cpu().numpy(), f"table {k} grad{c}")
            vm.zero_grad()
            for one in ones:
                one.zero_grad()
    torch.cuda.synchronize()
    for k, one in enumerate(ones):
        for c in range(3):
            assert_close(vm.table_rows(c)[k].cpu().numpy(), one.tt_cores[c][0].detach().cpu().numpy(),
                         f"table {k} core{c} after two steps")
            if optimizer == "adagrad":
                sv = torch.split(vm.optimizer_state[c][0], [p[c] for p in ps], dim=0)
                assert_close(sv[k].cpu().numpy(), one.optimizer_state[c][0].cpu().numpy(), f"table {k} state{c}")


@pytest.mark.parametrize("streams", [False, True])
def test_mixed_cardinality_tables(node, streams):
    """ttx_mixed.MixedTTEmbeddingBag: five tables of three different TT row shapes ->
    three table-batched groups, optionally on a HIP stream each; two fused-SGD steps leave every table's cores
    where a TTEmbeddingBag of its own leaves them, and the outputs agree."""
    import tt_embeddings_ops as ops
    import ttx_mixed

    D, q, r, B = 64, [4, 4, 4], [16, 16], 128
    Es = [9000, 60000, 8000, 900000, 50000]
    ps = [[20, 22, 25], [40, 40, 40], [20, 22, 25], [100, 100, 100], [40, 40, 40]]
    kw = dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, weight_dist="uniform", device=DEV)
    mm = ttx_mixed.MixedTTEmbeddingBag(Es, D, r, ps, q, include_last_offset=False, streams=streams, **kw)
    assert mm.group_tables == [[0, 2], [1, 4], [3]]
    ones = []
    for k in range(len(Es)):
        g = next(i for i, tb in enumerate(mm.group_tables) if k in tb)
        j = mm.group_tables[g].index(k)
        one = ops.TTEmbeddingBag(Es[k], D, r, ps[k], q, use_cache=False, include_last_offset=False, **kw)
        with torch.no_grad():
            for dst, src in zip(one.tt_cores, mm.groups[g].tt_cores):
                dst.copy_(src[j:j + 1])
        ones.append((one, g, j))
    rs = np.random.RandomState(11)
    grads = [t((rs.rand(B, D) * 0.1).astype(np.float32)) for _ in Es]
    for step in range(2):
        idx, off = [], []
        for e in Es:
            lens = rs.randint(0, 12, size=B)
            off.append(t(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)))
            idx.append(t(rs.randint(0, e, size=int(lens.sum())).astype(np.int64)))
        outs = mm(idx, off)
        torch.autograd.backward(outs, grads)
        for k, (one, g, j) in enumerate(ones):
            ref = one(idx[k], off[k])
            assert_close(outs[k].detach().cpu().numpy(), ref.detach().cpu().numpy(), f"step {step} table {k} forward")
            ref.backward(grads[k])
    torch.cuda.synchronize()
    for k, (one, g, j) in enumerate(ones):
        for c in range(3):
            assert_close(mm.groups[g].tt_cores[c][j].detach().cpu().numpy(), one.tt_cores[c][0].detach().cpu().numpy(),
                         f"table {k} core{c} after two SGD steps")


@pytest.mark.parametrize("fused", [False, True, "pad_q"])
def test_mixed_ranks_one_graph(node, fused):
    """f4: tables of different TT ranks AND factorings behind one MixedTTEmbeddingBag -- specialised (r = 32, 16) and generic
    (r = 13 / 12, q = [2, 4, 8]) launch sets, one per shape group, each on a HIP stream of its own and captured into ONE
    hipGraph as parallel branches.  The replayed round leaves every table's cores where one TTEmbeddingBag per table, stepped
    eagerly, leaves them; the eager forward agrees too."""
    import tt_embeddings_ops as ops
    import ttx_graph
    import ttx_mixed

    if node == "python":
        pytest.skip("graph capture of the module needs the C++ node")
    D, B, Lp = 64, 64, 6
    Es = [9000, 60000, 8000, 900000, 50000, 64000]
    ps = [[20, 22, 25], [40, 40, 40], [20, 22, 25], [100, 100, 100], [40, 40, 40], [40, 40, 40]]
    ranks = [[32, 32], [16, 16], [32, 32], [13, 12], [16, 16], [16, 16]]
    qs = [[4, 4, 4], [4, 4, 4], [4, 4, 4], [4, 4, 4], [4, 4, 4], [2, 4, 8]]
    kw = dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, weight_dist="uniform", device=DEV)
    pad_q = fused == "pad_q"
    fused = bool(fused)
    mm = ttx_mixed.MixedTTEmbeddingBag(Es, D, ranks, ps, qs, include_last_offset=False, streams=True, fused=fused, pad_ranks=True,
                                       pad_q=pad_q, **kw)
    # fused: ONE batched lookup per factoring q -- the five q = [4,4,4] tables of ranks 32 / 16 / [13,12] ride together, the smaller
    # ranks zero-padded to 32 -- and one for q = [2,4,8]; not fused: a group per (p, q, ranks); pad_q (round 4): ONE batched lookup
    # for all six, over q = [4,4,8] -- every table's cores zero-padded to it, its 64 values gathered out of the 128-value rows
    assert len(mm.groups) == (1 if pad_q else 2 if fused else 4) and sorted(sum(mm.group_tables, [])) == list(range(6))
    if pad_q:
        assert mm.groups[0].tt_q_shapes == [4, 4, 8] and mm.groups[0].out_dim == D
    ones = []
    for k in range(len(Es)):
        g = next(i for i, tb in enumerate(mm.group_tables) if k in tb)
        j = mm.group_tables[g].index(k)
        one = ops.TTEmbeddingBag(Es[k], D, ranks[k], ps[k], qs[k], use_cache=False, include_last_offset=False, **kw)
        rows = [mm.groups[g].table_core(j, c) for c in range(3)] if fused else [mm.groups[g].tt_cores[c][j] for c in range(3)]
        with torch.no_grad():
            for dst, src in zip(one.tt_cores, rows):
                dst.copy_(src.reshape(dst.shape))
        ones.append((one, g, j))
    rs = np.random.RandomState(12)
    grads = [t((rs.rand(B, D) * 0.1).astype(np.float32)) for _ in Es]
    reqs = []
    for _ in range(3):
        reqs.append(([t(rs.randint(0, e, size=B * Lp).astype(np.int64)) for e in Es],
                     [t(np.arange(0, B * Lp, Lp, dtype=np.int64)) for _ in Es]))
    outs = mm(*reqs[0])
    for k, (one, g, j) in enumerate(ones):
        assert_close(outs[k].detach().cpu().numpy(), one(*[x[k] for x in reqs[0]]).detach().cpu().numpy(), f"table {k} forward")
    rnd = ttx_graph.GraphedRound(lambda idx, off: torch.autograd.backward(mm(idx, off), grads), reqs, warmup=0)
    rnd.replay()
    for idx, off in reqs:
        for k, (one, g, j) in enumerate(ones):
            one(idx[k], off[k]).backward(grads[k])
    torch.cuda.synchronize()
    for k, (one, g, j) in enumerate(ones):
        for c in range(3):
            got = mm.groups[g].table_core(j, c) if fused else mm.groups[g].tt_cores[c][j].detach()
            assert_close(got.cpu().numpy().reshape(one.tt_cores[c].shape), one.tt_cores[c].detach().cpu().numpy(),
                         f"table {k} core{c} after the captured round")


def test_graphed_round_equals_eager_steps(node):
    """ttx_graph.GraphedRound: replaying a captured round of fused-SGD steps leaves the cores exactly where the
    same steps run eagerly leave them (the kernels are deterministic, so bit-identical)"""
    import tt_embeddings_ops as ops
    import ttx_graph

    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B = 20 * 22 * 25, 64, 96

    def fresh():
        m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=True,
                               cache_size=16, hashtbl_size=1 << 17, weight_dist="uniform", device=DEV)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, G.make_cores(31, 1, p, q, r, "signed")):
                dst.copy_(t(src))
        return m

    batches = [tuple(t(a) for a in G.make_bags(300 + k, B, E_, 6, 2, 1)) for k in range(4)]
    grad = t(G.make_grad(301, 1, B, D)[0])
    me, mg = fresh(), fresh()
    step_e = lambda i, o: me(i, o).backward(grad)  # noqa: E731
    step_g = lambda i, o: mg(i, o).backward(grad)  # noqa: E731
    before = [c.detach().clone() for c in mg.tt_cores]
    rnd = ttx_graph.GraphedRound(step_g, batches, warmup=2)
    with torch.no_grad():  # undo warm-up and capture-time side effects: same starting point as the eager model
        for c, b in zip(mg.tt_cores, before):
            c.copy_(b)
        mg.hashtbl.fill_(-1)
        mg.cache_freq.zero_()
    for _ in range(3):
        rnd.replay()
        for b in batches:
            step_e(*b)
    torch.cuda.synchronize()
    for a, b in zip(me.tt_cores, mg.tt_cores):
        assert torch.equal(a, b)
    # the frequency table counted the same keys the same number of times (slot assignment of keys that collide
    # depends on arrival order, as in the reference)
    def table(m):
        k, f = m.hashtbl.cpu().numpy(), m.cache_freq.cpu().numpy()
        return sorted(zip(k[k >= 0].tolist(), f[k >= 0].tolist()))
    assert table(me) == table(mg)


@pytest.mark.parametrize("tables,B,cap_extra", [(1, 64, 500), (3, 40, 7), (4, 600, 20000)])
def test_device_side_lookup_count(node, tables, B, cap_extra):
    """forward(indices, offsets, n_dev=): `indices` is a fixed-capacity buffer whose first n_dev entries are the batch (the rest:
    arbitrary valid indices that belong to no bag), the count stays on the device (ttx_lookup_prologue_n).  Output and the cores
    after a fused-SGD / Adagrad step -- and the dense gradients -- equal the plain call on the sliced batch BIT FOR BIT: the same
    plan, the same kernels on the same lookups (one-launch prologue, multi-launch plan and the large-batch kernels)."""
    import tt_embeddings_ops as ops

    if node == "python":
        pytest.skip("the device-side count is the C++ node's route (the python route reads it back)")
    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D = int(np.prod(p)), 64
    idx, off = G.make_bags(17 + B, B, E_, 9, 4, tables)
    n = idx.size
    rs = np.random.RandomState(3)
    buf = np.concatenate([idx, rs.randint(0, E_, size=cap_extra)]).astype(np.int64)
    grad = t(G.make_grad(18, tables, B, D))
    for kw in (dict(sparse=False), dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR),
               dict(sparse=True, optimizer=ops.OptimType.EXACT_ADAGRAD, learning_rate=LR, eps=EPS)):
        torch.manual_seed(1)
        a = ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, use_cache=False, weight_dist="uniform", device=DEV, **kw)
        torch.manual_seed(1)
        b = ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, use_cache=False, weight_dist="uniform", device=DEV, **kw)
        oa = a(t(buf), t(off), n_dev=torch.tensor([n], dtype=torch.int32, device=DEV))
        ob = b(t(idx), t(off))
        assert torch.equal(oa, ob), f"forward differs ({kw})"
        oa.backward(grad)
        ob.backward(grad)
        for k in range(3):
            if not kw["sparse"]:
                assert torch.equal(a.tt_cores[k].grad, b.tt_cores[k].grad), f"dense grad {k} differs"
            else:
                assert torch.equal(a.tt_cores[k].detach(), b.tt_cores[k].detach()), f"core {k} differs ({kw})"


def test_device_side_lookup_count_of_zero_and_refused_forms(node):
    """Round 6 (advisor): n_dev == 0 -- every bag empty, what a rank of the compact sharded path sees when it receives only empty
    bags -- plans, contracts and trains nothing: zero output, cores untouched (fused SGD), zero dense gradients; and
    per_sample_weights / include_last_offset=False are refused with n_dev instead of being silently dropped."""
    import tt_embeddings_ops as ops

    if node == "python":
        pytest.skip("the device-side count is the C++ node's route")
    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B, tables = int(np.prod(p)), 64, 32, 2
    buf = t(np.random.RandomState(5).randint(0, E_, size=500).astype(np.int64))
    off = torch.zeros(tables * B + 1, dtype=torch.int64, device=DEV)
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    grad = t(G.make_grad(19, tables, B, D))
    for kw in (dict(sparse=False), dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)):
        m = ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, use_cache=False, weight_dist="uniform", device=DEV, **kw)
        before = [c.detach().clone() for c in m.tt_cores]
        out = m(buf, off, n_dev=zero)
        assert out.shape == (tables, B, D) and not out.any()
        out.backward(grad)
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(m.tt_cores[k].detach(), before[k])
            if not kw["sparse"]:
                assert m.tt_cores[k].grad is not None and not m.tt_cores[k].grad.any()
        with pytest.raises(NotImplementedError):
            m(buf, off, n_dev=zero, per_sample_weights=torch.ones(500, device=DEV))
    m = ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, use_cache=False, weight_dist="uniform", device=DEV,
                                       include_last_offset=False)
    with pytest.raises(ValueError):
        m(buf, off[:-1], n_dev=zero)


def test_max_pooling_truncates_at_one_rank_as_it_does_at_many(node):
    """Round 4 advisor: forward(.., max_pooling=L) was ignored at world size 1 (bags longer than L were summed whole) and truncated at
    W > 1.  One rule everywhere: the first L lookups of every bag, the padding weighted zero -- against the table-batched module on
    the truncated bags, forward and one fused-SGD step; and pending planned batches are refused on this route too."""
    import tt_embeddings_ops as ops
    import ttx_sharded

    if node == "python":
        pytest.skip("zero-weight padding goes through per_sample_weights of the C++ node")
    p, q, r = [8, 9, 10], [4, 4, 4], [16, 16]
    E_, D, B, NT, Lmax = 720, 64, 24, 2, 5
    kw = dict(tt_p_shapes=p, tt_q_shapes=q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False,
              weight_dist="uniform", device=DEV)
    torch.manual_seed(3)
    sh = ttx_sharded.ShardedTableBatchedTTEmbeddingBag(NT, E_, D, r, **kw)
    assert sh.world == 1
    ref = ops.TableBatchedTTEmbeddingBag(NT, E_, D, r, **kw)
    with torch.no_grad():
        for x, y in zip(ref.tt_cores, sh.local.tt_cores):
            x.copy_(y)
    rs = np.random.RandomState(9)
    lens = rs.randint(0, 2 * Lmax + 1, size=NT * B)
    assert lens.max() > Lmax
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = rs.randint(0, E_, size=int(lens.sum())).astype(np.int64)
    keep = np.concatenate([np.arange(off[b], off[b] + min(lens[b], Lmax)) for b in range(NT * B)]).astype(np.int64)
    toff = np.concatenate([[0], np.cumsum(np.minimum(lens, Lmax))]).astype(np.int64)
    grad = t(G.make_grad(5, NT, B, D))
    out = sh(t(idx), t(off), max_pooling=Lmax)
    want = ref(t(idx[keep]), t(toff))
    assert_close(out.detach().cpu().numpy(), want.detach().cpu().numpy(), "truncated forward")
    out.backward(grad)
    want.backward(grad)
    for k in range(3):
        assert_close(sh.local.tt_cores[k].detach().cpu().numpy(), ref.tt_cores[k].detach().cpu().numpy(), f"core {k} after the step")
    sh._planned = {(1, 2): None}
    with pytest.raises(RuntimeError, match="planned ahead"):
        sh(t(idx), t(off), max_pooling=Lmax)


def test_direct_rccl_exchange_one_rank():
    """ttx_sharded.DirectExchange (ncclAllToAll on the current stream, csrc/ttx_torch.cpp) against the
    torch.distributed route, eagerly and captured in a hipGraph -- with the one rank this box has; own process,
    because the process group cannot be torn down cleanly on this stack"""
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(__file__), "_direct_exchange_worker.py")
    res = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=240)
    assert "DIRECT-EXCHANGE-OK" in res.stdout and "RAGGED-PADDED-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.parametrize("use_cache", [False, True])
def test_prefetched_prologue_is_bit_identical_eager_and_captured(node, use_cache):
    """module.prefetch(): the lookup prologue of the NEXT batch on a side stream while this step's backward runs.
    Same kernels on the same inputs -> outputs, cores and frequency table bit-identical to the plain sequence, eagerly
    and as one captured round (ttx_graph.pipelined_round inside a GraphedRound)."""
    import tt_embeddings_ops as ops
    import ttx_graph

    if node == "python":
        m = ops.TTEmbeddingBag(11000, 64, [16, 16], [20, 22, 25], [4, 4, 4], use_cache=False, weight_dist="uniform", device=DEV)
        assert m.prefetch(torch.zeros(4, dtype=torch.int64, device=DEV), torch.tensor([0, 4], device=DEV)) is False
        return
    p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
    E_, D, B, Lp = 11_000_000, 64, 512, 20
    kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform", device=DEV,
              sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=use_cache, cache_size=1024, hashtbl_size=1 << 23)  # (sparse table: no key is dropped, whatever the insert order)
    torch.manual_seed(5)
    a, b, c = ops.TTEmbeddingBag(**kw), ops.TTEmbeddingBag(**kw), ops.TTEmbeddingBag(**kw)
    with torch.no_grad():
        for other in (b, c):
            for x, y in zip(other.tt_cores, a.tt_cores):
                x.copy_(y)
    reqs = [(t(i), t(o)) for i, o in G.make_requests(31, 6, B, 1, Lp, E_)]
    grad = t(G.make_grad(32, 1, B, D)[0])
    outs_init = [x.detach().clone() for x in a.tt_cores]
    outs_a, outs_b = [], []
    for k, (i, o) in enumerate(reqs):  # plain
        out = a(i, o)
        outs_a.append(out.detach().clone())
        out.backward(grad)
    for k, (i, o) in enumerate(reqs):  # prefetched, eager
        out = b(i, o)
        if k + 1 < len(reqs):
            assert b.prefetch(*reqs[k + 1]) is True
        outs_b.append(out.detach().clone())
        out.backward(grad)
    torch.cuda.synchronize()
    for x, y in zip(outs_a, outs_b):
        assert torch.equal(x, y), "prefetched forward differs"
    for x, y in zip(a.tt_cores, b.tt_cores):
        assert torch.equal(x, y), "cores differ after the prefetched steps"
    c.prefetch_stream()  # (the side stream exists before the capture begins)
    rnd = ttx_graph.GraphedRound(ttx_graph.pipelined_round(c, reqs, lambda out, k: out.backward(grad)), [()], warmup=0)
    torch.cuda.synchronize()  # (warmup=0 and a capture does not execute: c still holds the initial cores)
    rnd.replay()
    torch.cuda.synchronize()
    for x, y in zip(a.tt_cores, c.tt_cores):
        assert torch.equal(x, y), "cores differ after the captured pipelined round"
    # ... and all six prologues in ONE launch up front (prefetch_many), eagerly and as a captured round
    d, e = ops.TTEmbeddingBag(**kw), ops.TTEmbeddingBag(**kw)
    with torch.no_grad():
        for other in (d, e):
            for x, y in zip(other.tt_cores, outs_init):
                x.copy_(y)
    assert d.prefetch_many(reqs) is True
    for k, (i, o) in enumerate(reqs):
        out = d(i, o)
        assert torch.equal(out.detach(), outs_a[k]), "forward after prefetch_many differs"
        out.backward(grad)
    e.prefetch_stream()
    rnd2 = ttx_graph.GraphedRound(ttx_graph.planned_round(e, reqs, lambda out, k: out.backward(grad)), [()], warmup=0)
    torch.cuda.synchronize()
    rnd2.replay()
    torch.cuda.synchronize()
    for x, y, z in zip(a.tt_cores, d.tt_cores, e.tt_cores):
        assert torch.equal(x, y) and torch.equal(x, z), "cores differ after the planned rounds"
    if use_cache:
        def table(m):
            k_, f_ = m.hashtbl.cpu().numpy(), m.cache_freq.cpu().numpy()
            return sorted(zip(k_[k_ >= 0].tolist(), f_[k_ >= 0].tolist()))
        assert table(a) == table(b) == table(c) == table(d) == table(e), "frequency tables differ"


def test_prefetch_of_a_batch_that_changed_is_not_used(node):
    """a prefetched prologue belongs to the tensor OBJECTS it was made for, as they were: a batch written to after its
    prefetch, or another tensor, runs the prologue in line (no stale plan)"""
    import tt_embeddings_ops as ops

    if node == "python":
        return
    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    m = ops.TTEmbeddingBag(11000, 64, r, p, q, sparse=False, use_cache=False, weight_dist="uniform", device=DEV)
    idx, off = (t(a) for a in G.make_bags(3, 300, 11000, 6, 2, 1))
    ref = m(idx, off).detach().clone()
    assert m.prefetch(idx, off) is True
    idx2 = idx.clone()
    assert torch.equal(m(idx2, off).detach(), ref) and len(m._prefetched) == 1, "another tensor object: in-line prologue"
    idx.add_(1).remainder_(11000)  # written to after the prefetch
    out = m(idx, off).detach()
    assert len(m._prefetched) == 0, "the stale entry must be dropped"
    fresh = ops.TTEmbeddingBag(11000, 64, r, p, q, sparse=False, use_cache=False, weight_dist="uniform", device=DEV)
    with torch.no_grad():
        for x, y in zip(fresh.tt_cores, m.tt_cores):
            x.copy_(y)
    assert torch.equal(out, fresh(idx, off).detach()), "a modified batch must be planned again"


def _cache_live_pair(ops, p, q, r, E_, D, B, Lp, optimizer, n_req=5, cache_size=512):
    """two modules with the same cores whose caches went live on the same warm-up stream, and a round of requests"""
    kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform", device=DEV,
              sparse=optimizer is not None, optimizer=optimizer or ops.OptimType.SGD, learning_rate=0.05, use_cache=True,
              cache_size=cache_size, hashtbl_size=1 << 20)
    torch.manual_seed(11)
    a, b = ops.TTEmbeddingBag(**kw), ops.TTEmbeddingBag(**kw)
    warm = [(t(i), t(o)) for i, o in G.make_requests(70, 3, B, 1, Lp, E_, alpha=1.2)]
    with torch.no_grad():
        for i, o in warm:
            a(i, o)
    a.cache_populate()
    # (b takes a's table and cache as they are: two tables filled by the same stream may seat colliding keys in a different
    #  order -- as in the reference -- and then break frequency ties at the cache's edge differently)
    b.load_state_dict(a.state_dict())
    b.warmup = False
    # Keys the comparison "planned ahead == in line" cannot be made on: a CACHED key that sits behind its home slot with an
    # empty slot in front of it (its home's first tenant was evicted by the populate).  The reference's insert claims the
    # first empty-or-matching slot of the probe sequence (hashtbl_cuda_utils.cuh:102-133), so the next time such a key is
    # counted it gets a SECOND seat in front of its cached one and is a miss from then on -- unless another key has taken
    # that empty slot first.  Which of the two happens depends on the order in which the batches are counted: one after the
    # other in line, all of a planned round at once (prefetch_many).  Either outcome is a legal reference behaviour (there the
    # order of two racing inserts is the hardware's); the requests simply leave those few keys out.
    keys, state = a.hashtbl.cpu().numpy(), a.cache_state.cpu().numpy()
    H = keys.size
    prone = set()
    for s_ in np.nonzero(state >= 0)[0]:
        k = int(keys[s_])
        h = O.hash64(k, H)
        c = h
        while c != s_:
            if keys[c] == -1:
                prone.add(k)
                break
            c = (c + 1) % H
    reqs = []
    for i, o in G.make_requests(71, n_req, B, 1, Lp, E_, alpha=1.2):
        if prone:
            safe = next(v for v in range(E_) if v not in prone)
            i = np.where(np.isin(i, list(prone)), safe, i)
        reqs.append((t(i), t(o)))
    return a, b, reqs


@pytest.mark.parametrize("shape", ["single-launch", "small", "general", "q0=2"])
def test_cache_live_round_planned_ahead_equals_the_in_line_prologue(node, shape):
    """prefetch_many() with a LIVE cache (ttx_lookup_prologue_cached_multi): per batch the same partition (misses first
    in index order, hits behind them reversed), cache locations, split point and -- through the forward it drives --
    plan as the in-line route (ttx_preprocess_indices_async + ttx_plan_build_n)"""
    import tt_embeddings as E
    import tt_embeddings_ops as ops

    if node == "python":
        return
    p, q, r, E_, D, B, Lp = {"single-launch": ([200, 220, 250], [4, 4, 4], [32, 32], 11_000_000, 64, 512, 20),
                             "small": ([20, 22, 25], [4, 4, 4], [16, 16], 11_000, 64, 96, 5),
                             "general": ([300, 20, 20], [4, 4, 4], [8, 8], 120_000, 64, 128, 12),
                             "q0=2": ([20, 22, 25], [2, 4, 4], [16, 16], 11_000, 32, 96, 5)}[shape]
    a, b, reqs = _cache_live_pair(ops, p, q, r, E_, D, B, Lp, None)
    assert b.prefetch_many(reqs) is True and len(b._prefetched) == len(reqs)
    hits = 0
    for i, o in reqs:
        key = (id(i), id(o))
        tableidx, pcol, prow, ploc, n_tt, plan = b._prefetched[key][2]
        off = torch.cat([o, o.new_full((1,), i.numel())]) if not a.include_last_offset else o
        ecol, erow, etab, e_ntt, eloc = E.preprocess_indices_sync(i, off, 1, False, a.hashtbl, a.cache_state, a.cache_freq)
        assert int(n_tt.item()) == e_ntt
        assert torch.equal(pcol, ecol) and torch.equal(prow, erow) and torch.equal(ploc, eloc)
        assert torch.equal(tableidx, torch.zeros_like(tableidx))
        hits += i.numel() - e_ntt
        out_b = b(i, o)   # planned (consumes the entry)
        a_out = a(i, o)   # in line (counts the batch a second time in a's table: lookups do not depend on the counts)
        assert torch.equal(out_b.detach(), a_out.detach()), "forward through the planned-ahead plan differs"
    assert hits > 0 and len(b._prefetched) == 0


@pytest.mark.parametrize("det", [None, True])
@pytest.mark.parametrize("optim", ["sgd", "adagrad"])
def test_cache_live_training_round_planned_ahead_eager_and_captured(node, optim, det):
    """a round of cache-live training steps with the prologues planned ahead, eagerly and replayed from a hipGraph,
    against the plain sequence: first output bit-identical, the rest to rounding (the cache rows' fused SGD updates are
    float atomics whose order is not fixed, here as in the reference); the frequency table counts the same.  With
    Adagrad the cache rows' step sizes depend on which lookup of a row arrives first (cu:1735-1795, `old` of the
    atomic) -- two plain runs differ in the second step already -- so only the first output and the table are held.
    det = True (round 6, `deterministic_cache_update`): the cache rows are updated without atomics, in index order within a row --
    then EVERY output of the planned-ahead round and of the captured replay, the cores, the cache rows and (Adagrad) the cache's
    optimizer state are BIT-identical to the plain sequence, for both optimizers."""
    import tt_embeddings_ops as ops
    import ttx_graph

    if node == "python":
        return
    p, q, r, E_, D, B, Lp = [200, 220, 250], [4, 4, 4], [32, 32], 11_000_000, 64, 512, 20
    optimizer = ops.OptimType.SGD if optim == "sgd" else ops.OptimType.EXACT_ADAGRAD
    a, b, reqs = _cache_live_pair(ops, p, q, r, E_, D, B, Lp, optimizer, n_req=4)
    a.deterministic_cache_update = b.deterministic_cache_update = det
    c = ops.TTEmbeddingBag(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform",
                           device=DEV, sparse=True, optimizer=optimizer, learning_rate=0.05, use_cache=True, cache_size=512,
                           hashtbl_size=1 << 20, deterministic_cache_update=det)
    c.load_state_dict(a.state_dict())
    c.warmup = False
    grad = t(G.make_grad(72, 1, B, D)[0])
    outs_a = []
    for i, o in reqs:
        out = a(i, o)
        outs_a.append(out.detach().clone())
        out.backward(grad)
    assert b.prefetch_many(reqs) is True
    for k, (i, o) in enumerate(reqs):
        out = b(i, o)
        if k == 0:
            assert torch.equal(out.detach(), outs_a[0])
        if det:
            assert torch.equal(out.detach(), outs_a[k]), f"step {k}: the planned-ahead round must equal the plain sequence bit for bit"
        elif optim == "sgd":
            assert torch.allclose(out.detach(), outs_a[k], rtol=2e-5, atol=2e-6), f"step {k}"
        assert bool(torch.isfinite(out).all())
        out.backward(grad)
    c.prefetch_stream()
    rnd = ttx_graph.GraphedRound(ttx_graph.planned_round(c, reqs, lambda out, k: out.backward(grad)), [()], warmup=0)
    torch.cuda.synchronize()
    rnd.replay()
    torch.cuda.synchronize()
    if det:
        for x, y, z in zip(a.tt_cores, b.tt_cores, c.tt_cores):
            assert torch.equal(x, y) and torch.equal(x, z)
        assert torch.equal(a.cache_weight, b.cache_weight) and torch.equal(a.cache_weight, c.cache_weight)
        if optim != "sgd":
            assert torch.equal(a.cache_optimizer_state, b.cache_optimizer_state) and torch.equal(a.cache_optimizer_state, c.cache_optimizer_state)
    elif optim == "sgd":
        for x, y, z in zip(a.tt_cores, b.tt_cores, c.tt_cores):
            assert torch.allclose(x, y, rtol=2e-5, atol=2e-6) and torch.allclose(x, z, rtol=2e-5, atol=2e-6)
        assert torch.allclose(a.cache_weight, b.cache_weight, rtol=2e-5, atol=2e-6)
        assert torch.allclose(a.cache_weight, c.cache_weight, rtol=2e-5, atol=2e-6)

    def table(m):
        k_, f_ = m.hashtbl.cpu().numpy(), m.cache_freq.cpu().numpy()
        return sorted(zip(k_[k_ >= 0].tolist(), f_[k_ >= 0].tolist()))
    assert table(a) == table(b) == table(c)


def test_cache_populate_drops_what_was_planned_ahead(node):
    """a batch planned ahead was split into hits and misses by the cache contents of that moment: cache_populate() and
    reset_cache() void it, a prologue planned while the cache was warming up is not used once it is live"""
    import tt_embeddings_ops as ops

    if node == "python":
        return
    a, b, reqs = _cache_live_pair(ops, [20, 22, 25], [4, 4, 4], [16, 16], 11_000, 64, 96, 5, None, n_req=2)
    assert b.prefetch_many(reqs) is True and len(b._prefetched) == 2
    b.cache_populate()
    assert len(b._prefetched) == 0
    for (i, o) in reqs:  # (b's second populate may cache other rows than a's first: same values to rounding)
        assert torch.allclose(a(i, o).detach(), b(i, o).detach(), rtol=1e-5, atol=1e-6)
    b.reset_cache()  # warming up again
    assert len(b._prefetched) == 0 and b.prefetch_many(reqs[:1]) is True
    b.warmup = False  # (an entry planned for the warming-up cache is not of the kind the live route takes)
    assert torch.allclose(b(*reqs[0]).detach(), a(*reqs[0]).detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("live", [False, True])
def test_more_batches_than_one_prologue_launch_holds(node, live):
    """prefetch_many with 18 batches: the multi-batch prologues take 16 batches per launch (the pointer table of
    ProBatch) -- the 17th and 18th go through a second one; every batch's forward equals the in-line one"""
    import tt_embeddings_ops as ops

    if node == "python":
        return
    p, q, r, E_, D, B, Lp = [20, 22, 25], [4, 4, 4], [16, 16], 11_000, 64, 300, 6
    if live:
        a, b, reqs = _cache_live_pair(ops, p, q, r, E_, D, B, Lp, None, n_req=18, cache_size=256)
    else:
        kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform",
                  device=DEV, sparse=False, use_cache=True, cache_size=256, hashtbl_size=1 << 20)
        torch.manual_seed(3)
        a, b = ops.TTEmbeddingBag(**kw), ops.TTEmbeddingBag(**kw)
        b.load_state_dict(a.state_dict())
        reqs = [(t(i), t(o)) for i, o in G.make_requests(75, 18, B, 1, Lp, E_, alpha=1.2)]
    assert b.prefetch_many(reqs) is True and len(b._prefetched) == 18
    for i, o in reqs:
        assert torch.equal(b(i, o).detach(), a(i, o).detach())
    assert len(b._prefetched) == 0


@pytest.mark.parametrize("det", [None, True])  # (round 6: True = the atomic-free cache-row update, every lookup its own scaled gradient row)
@pytest.mark.parametrize("optim", ["dense", "sgd", "adagrad"])
def test_per_sample_weights_with_a_live_cache(node, optim, det):
    """nn.EmbeddingBag's per_sample_weights while the row cache is live (the part the reference has no
    counterpart for at all): against torch's embedding_bag + autograd on the table the module serves at that moment --
    the TT rows with the cached rows laid over them.  Forward, the weights' own gradient, the dense gradients of the cores
    (misses only) and of the cache rows (hits only); with the fused optimizers the cache rows' and cores' steps."""
    import tt_embeddings as E
    import tt_embeddings_ops as ops

    if node == "python":
        return
    p, q, r, E_, D, B, Lp = [20, 22, 25], [4, 4, 4], [16, 16], 11_000, 64, 96, 6
    optimizer = {"dense": None, "sgd": ops.OptimType.SGD, "adagrad": ops.OptimType.EXACT_ADAGRAD}[optim]
    a, _, reqs = _cache_live_pair(ops, p, q, r, E_, D, B, Lp, optimizer, n_req=1, cache_size=200)
    a.deterministic_cache_update = det
    idx, off = reqs[0]
    rs = np.random.RandomState(4)
    psw = t((rs.rand(idx.numel()) * 2 - 0.5).astype(np.float32))
    d_out = t(G.make_grad(9, 1, B, D)[0])
    # the table the module serves now, as a function of the cores and the cache rows
    keys, state = a.hashtbl.cpu().numpy(), a.cache_state.cpu().numpy()
    slot_ok = (keys >= 0) & (state >= 0)
    ck, crow = torch.from_numpy(keys[slot_ok]).to(DEV), torch.from_numpy(state[slot_ok].astype(np.int64)).to(DEV)
    cached = torch.zeros(E_, dtype=torch.bool, device=DEV)
    cached[ck] = True
    hit = cached[idx]
    assert 0 < int(hit.sum()) < idx.numel(), "the batch must hold hits and misses"
    ref_cores = [c.detach().clone().requires_grad_(True) for c in a.tt_cores]
    ref_cw = a.cache_weight.detach().clone().requires_grad_(True)
    full = ops.tt_matrix_to_full(p, q, [1] + r + [1], ref_cores, [1, 0, 2, 3])
    table = full.index_put((ck,), ref_cw[crow])  # cached rows laid over the TT rows
    w_ref = psw.clone().requires_grad_(True)
    ref = torch.nn.functional.embedding_bag(idx, table, off, mode="sum", per_sample_weights=w_ref, include_last_offset=True)
    ref.backward(d_out)
    cores0 = [c.detach().clone() for c in a.tt_cores]
    cw0 = a.cache_weight.detach().clone()
    w = psw.clone().requires_grad_(True)
    out = a(idx, off, per_sample_weights=w)
    assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), "weighted forward, cache live")
    out.backward(d_out)
    assert w.grad is not None
    assert_close(w.grad.cpu().numpy(), w_ref.grad.cpu().numpy(), "gradient of per_sample_weights, cache live")
    if optim == "dense":
        for k in range(3):
            assert_close(a.tt_cores[k].grad.cpu().numpy(), ref_cores[k].grad.cpu().numpy(), f"core {k} gradient (misses)")
        assert_close(a.cache_weight.grad.cpu().numpy(), ref_cw.grad.cpu().numpy(), "cache row gradient (hits)")
    elif optim == "sgd":
        for k in range(3):
            assert_close(a.tt_cores[k].detach().cpu().numpy(), (cores0[k] - 0.05 * ref_cores[k].grad).cpu().numpy(), f"core {k} after SGD")
        assert_close(a.cache_weight.detach().cpu().numpy(), (cw0 - 0.05 * ref_cw.grad).cpu().numpy(), "cache rows after SGD",
                     rtol=2e-5, atol_scale=4e-6)
    else:  # Adagrad: the rows that were not hit stay, the rows that were hit move against their gradient
        moved = (a.cache_weight.detach() - cw0).abs().sum(dim=1) > 0
        touched = ref_cw.grad.abs().sum(dim=1) > 0
        assert torch.equal(moved, touched)
        step = a.cache_weight.detach() - cw0
        assert bool(((step * ref_cw.grad).sum(dim=1)[touched] < 0).all())
    # without a gradient for the weights, and an unweighted call afterwards, the module keeps working
    a(idx, off, per_sample_weights=psw).backward(d_out)
    a(idx, off).backward(d_out)


def test_discarded_planned_batches_are_counted_once_and_the_module_copies(node):
    """A batch whose lookup prologue was planned ahead (prefetch / prefetch_many) has been counted into the frequency
    table at that point.  When the planned entry is then not used -- cache_populate() in between, more than eight single
    prefetches pending, per_sample_weights over a live cache -- the in-line prologue looks the indices up WITHOUT counting
    them a second time (the LFU ranking of the next populate stays what the plain sequence gives).  And a module that has
    prefetched deep-copies and pickles (the side stream and the planned buffers stay behind)."""
    import copy
    import io

    import tt_embeddings_ops as ops

    if node == "python":
        pytest.skip("prefetch needs the C++ node")
    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B, Lp = 11000, 64, 64, 5
    kw = dict(num_embeddings=E_, embedding_dim=D, tt_ranks=r, tt_p_shapes=p, tt_q_shapes=q, weight_dist="uniform", device=DEV,
              sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=True, cache_size=256, hashtbl_size=1 << 16)
    torch.manual_seed(7)
    m = ops.TTEmbeddingBag(**kw)
    reqs = [(t(i), t(o)) for i, o in G.make_requests(41, 12, B, 1, Lp, E_)]
    nnz = B * Lp
    # (1) planned, then cache_populate() drops the plan: the batch is not counted again when it comes
    assert m.prefetch_many(reqs[:2]) is True
    assert int(m.cache_freq.sum()) == 2 * nnz
    m.cache_populate()  # (resets the counts of the rows it caches / evicts: compare against what it leaves)
    s0 = int(m.cache_freq.sum())
    for i, o in reqs[:2]:
        m(i, o)
    assert int(m.cache_freq.sum()) == s0, "a batch planned before cache_populate() was counted twice"
    m(*reqs[4])
    assert int(m.cache_freq.sum()) == s0 + nnz  # (a batch that was never planned is counted as ever)
    # (2) live cache, planned, then used with per_sample_weights (the planned partition carries no weights)
    assert m.prefetch_many(reqs[2:4]) is True
    assert int(m.cache_freq.sum()) == s0 + 3 * nnz
    w = torch.rand(nnz, device=DEV)
    m(reqs[2][0], reqs[2][1], per_sample_weights=w)
    m(reqs[3][0], reqs[3][1])
    assert int(m.cache_freq.sum()) == s0 + 3 * nnz, "a weighted batch over a planned entry was counted twice"
    # (3) nine single prefetches: the oldest is evicted, comes after all, and is counted once
    m.reset_cache()
    for i, o in reqs[:9]:
        assert m.prefetch(i, o) is True
    for i, o in reqs[:9]:
        m(i, o)
    torch.cuda.synchronize()
    assert int(m.cache_freq.sum()) == 9 * nnz, "an evicted planned batch was counted twice"
    # (4) the module copies and pickles after having prefetched
    assert m.prefetch(*reqs[9]) is True
    m2 = copy.deepcopy(m)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    out = m(*reqs[9])
    for other in (m2, m3):
        assert not getattr(other, "_prefetched", None)
        assert torch.equal(other(*reqs[9]), out)


@pytest.mark.parametrize("q,ranks,tables", [([8, 8, 8], [16, 16], 1), ([8, 4, 4], [32, 32], 3), ([6, 4, 8], [16, 24], 2),
                                             ([12, 4, 4], [16, 16], 1), ([16, 2, 4], [13, 12], 1),
                                             ([8, 8, 16], [16, 16], 1), ([8, 8, 12], [32, 32], 1), ([8, 8, 10], [24, 32], 2),
                                             ([8, 8, 16], [64, 64], 1),
                                             # q0 without an exact split (round 4): core 0 zero-padded to 8 / 12 / 16 slots
                                             ([5, 8, 8], [32, 32], 1), ([7, 4, 4], [16, 16], 2), ([10, 4, 8], [16, 24], 1),
                                             ([13, 2, 4], [13, 12], 1),
                                             # q1 beyond 8 (round 5: templates with q1 = 16 at ranks <= 32): the default factorings
                                             # of D = 720 / 800 / 960 / 1008
                                             ([8, 9, 10], [32, 32], 1), ([8, 10, 10], [16, 16], 2), ([8, 10, 12], [24, 32], 1),
                                             ([7, 12, 12], [32, 32], 1),
                                             # ... and a prime last factor (templates with q2 = 32): D = 816 / 912
                                             ([6, 8, 17], [32, 32], 1), ([6, 8, 19], [16, 16], 2), ([8, 9, 10], [64, 64], 1)])
def test_first_factor_beyond_four_runs_as_part_lookups(q, ranks, tables):
    """q0 > 4 (the reference's default factoring of D = 512 is [8, 8, 8]): core 0 [p0, q0, r1] IS [k p0, q0 / k, r1], every
    index becomes k part lookups whose rows are the k parts of the bag's output row (include/ttx.h "core-0 row split"), and the
    shape-specialised kernels (q0 <= 4) take the table.  Against the oracle on the ORIGINAL geometry and against the same
    module with the split switched off (generic kernels); dense gradients, fused SGD and Adagrad; ragged and empty bags."""
    import tt_embeddings_ops as ops

    p = [5, 6, 7]
    r = [1] + ranks + [1]
    E_, D, B = int(np.prod(p)), int(np.prod(q)), 50
    idx, off = G.make_bags(71, B, E_, 4, 3, tables)
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
             cores=G.make_cores(72, tables, p, q, r, "signed"), d_out=G.make_grad(73, tables, B, D))
    g = O.make_geom(tables, p, q, r)
    rowidx, tableidx = O.rowidx_from_offsets(off, tables)

    def run(split, **kw):
        m = module_for(c, **kw)
        assert m._split0 > 1, "the geometry is expected to split"
        if not split:
            m._split0 = 0
        out = m(t(idx), t(off))
        out.backward(t(c["d_out"]))
        return m, out.detach().cpu().numpy()

    # dense
    ms, outs = run(True, sparse=False)
    mg, outg = run(False, sparse=False)
    ref_out = O.tt_forward(g, B, D, idx, rowidx, tableidx, [x.copy() for x in c["cores"]])
    ref_g = O.tt_backward(g, O.OPTIM_DENSE, B, D, 0, 0, idx, rowidx, tableidx, c["d_out"], [x.copy() for x in c["cores"]])
    assert_close(outs, ref_out, f"split q={q} out vs oracle")
    assert_close(outs, outg, f"split q={q} out vs generic")
    for k in range(3):
        assert_close(ms.tt_cores[k].grad.cpu().numpy(), ref_g[k], f"split q={q} grad{k} vs oracle")
        assert_close(ms.tt_cores[k].grad.cpu().numpy(), mg.tt_cores[k].grad.cpu().numpy(), f"split q={q} grad{k} vs generic")
    # fused SGD / Adagrad
    ms, _ = run(True, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)
    cores = [x.copy() for x in c["cores"]]
    O.tt_backward(g, O.OPTIM_SGD, B, D, LR, 0, idx, rowidx, tableidx, c["d_out"], cores)
    for k in range(3):
        assert_close(ms.tt_cores[k].detach().cpu().numpy(), cores[k], f"split q={q} sgd core{k}")
    ms, _ = run(True, sparse=True, optimizer=ops.OptimType.EXACT_ADAGRAD, learning_rate=LR, eps=EPS)
    cores, state = [x.copy() for x in c["cores"]], [np.zeros_like(x) for x in c["cores"]]
    O.tt_backward(g, O.OPTIM_ADAGRAD, B, D, LR, EPS, idx, rowidx, tableidx, c["d_out"], cores, state)
    for k in range(3):
        assert_close(ms.optimizer_state[k].cpu().numpy(), state[k], f"split q={q} adagrad state{k}")
        assert_adagrad_close(ms.tt_cores[k].detach().cpu().numpy(), cores[k], ref_g[k], f"split q={q} adagrad core{k}")


@pytest.mark.parametrize("optimizer", ["SGD", "EXACT_ADAGRAD"])
def test_padded_first_factor_over_several_steps(node, optimizer):
    """q = [5, 8, 8] (the reference's default factoring of D = 320): the part lookups run on a zero-padded copy of core 0 (and of
    its optimizer state) that the module keeps -- refreshed from the Parameter when that was written, written back after every
    fused update.  Several steps, an in-place write and a load_state_dict between them, then a captured step: the module's own
    core 0 / state 0 must follow the same module on the generic kernels (split off) all the way."""
    import tt_embeddings_ops as ops
    import ttx_graph

    p, q, ranks = [6, 7, 8], [5, 8, 8], [16, 16]
    r = [1] + ranks + [1]
    E_, D, B, tables = int(np.prod(p)), int(np.prod(q)), 40, 2
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, cores=G.make_cores(82, tables, p, q, r, "signed"))
    opt = getattr(ops.OptimType, optimizer)
    a = module_for(c, sparse=True, optimizer=opt, learning_rate=LR, eps=EPS)
    b = module_for(c, sparse=True, optimizer=opt, learning_rate=LR, eps=EPS)
    assert a._pad0 == 8 and a._split0 == 2
    b._split0 = 0  # the generic kernels on the natural geometry
    batches = [G.make_bags(90 + k, B, E_, 4, 3, tables) for k in range(4)]
    grad = t(G.make_grad(83, tables, B, D))

    def same(what):
        for k in range(3):
            x, y = a.tt_cores[k].detach().cpu().numpy(), b.tt_cores[k].detach().cpu().numpy()
            if optimizer == "SGD":
                assert_close(x, y, f"{what}: core{k}")
                continue
            # Adagrad: the state (sum of squared gradients) at the gradient tolerance; the cores' update g / (sqrt(state) + eps)
            # amplifies the rounding of a near-zero gradient (util.assert_adagrad_close) -- a few entries may differ, by less
            # than one update
            assert_close(a.optimizer_state[k].cpu().numpy(), b.optimizer_state[k].cpu().numpy(), f"{what}: state{k}")
            bad = np.abs(x - y) > 1e-5 * np.abs(y) + 1e-6 * np.abs(y).max()
            assert bad.mean() < 0.05 and np.abs(x - y).max() < 6 * LR, f"{what}: core{k}: {int(bad.sum())}/{bad.size} differ"

    def outs_close(oa, ob, what):  # (Adagrad: the cores agree up to the ill-conditioned entries, see same())
        x, y = oa.detach().cpu().numpy(), ob.detach().cpu().numpy()
        if optimizer == "SGD":
            assert_close(x, y, what)
        else:
            np.testing.assert_allclose(x, y, rtol=2e-3, atol=2e-3 * float(np.abs(y).max()), err_msg=what)

    for k in range(2):
        oa, ob = a(t(batches[k][0]), t(batches[k][1])), b(t(batches[k][0]), t(batches[k][1]))
        assert oa.is_contiguous() and oa.shape == ob.shape
        outs_close(oa, ob, f"step {k} out")
        oa.backward(grad)
        ob.backward(grad)
        same(f"step {k}")
    with torch.no_grad():  # an in-place write of the Parameter: the padded copy must be refreshed
        for m in (a, b):
            m.tt_cores[0].mul_(0.5)
    sd = {k_: v.clone() for k_, v in b.state_dict().items()}
    for k in range(2, 4):
        oa, ob = a(t(batches[k][0]), t(batches[k][1])), b(t(batches[k][0]), t(batches[k][1]))
        outs_close(oa, ob, f"step {k} out (after the in-place write)")
        oa.backward(grad)
        ob.backward(grad)
        same(f"step {k}")
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    fixed = G.make_requests(95, 3, B, tables, 3, E_)  # (a captured step replays one shape)
    step = ttx_graph.GraphedStep(lambda i, o: a(i, o).backward(grad), (t(fixed[0][0]), t(fixed[0][1])), warmup=0)
    before = [x.detach().clone() for x in a.tt_cores]
    for i, o in fixed[1:]:
        step(t(i), t(o))
        b(t(i), t(o)).backward(grad)
    torch.cuda.synchronize()
    assert not torch.equal(before[0], a.tt_cores[0].detach()), "the captured step did not write core 0 back"
    same("captured steps")


@pytest.mark.parametrize("optimizer", ["SGD", "EXACT_ADAGRAD"])
def test_padded_first_factor_over_several_steps_tracks_the_oracle(node, optimizer):
    """The same life cycle against the ORACLE run step by step on the natural geometry (round 4 verdict: the multi-step logic of the
    padded copy -- refresh, write-back -- had only the module itself on the generic kernels behind it): q = [5, 8, 8], two tables,
    two steps, a write THROUGH `.data` (which does not move the Parameter's version counter: the round-4 copy went stale on it,
    round 4 advisor), two more steps, a load_state_dict, a step -- forward outputs, cores and Adagrad state after every step."""
    import tt_embeddings_ops as ops

    p, q, ranks = [6, 7, 8], [5, 8, 8], [16, 16]
    r = [1] + ranks + [1]
    E_, D, B, tables = int(np.prod(p)), int(np.prod(q)), 40, 2
    cores = G.make_cores(82, tables, p, q, r, "signed")
    c = dict(tables=tables, T=3, p=p, q=q, r=r, B=B, D=D, cores=cores)
    opt = getattr(ops.OptimType, optimizer)
    lr, eps = 0.05, 1e-3
    a = module_for(c, sparse=True, optimizer=opt, learning_rate=lr, eps=eps)
    assert a._pad0 == 8 and a._split0 == 2
    g = O.make_geom(tables, p, q, ranks)
    ref = [x.copy() for x in cores]
    state = [np.zeros_like(x) for x in cores]
    adagrad = optimizer != "SGD"

    def one(step):
        idx, off = G.make_bags(90 + step, B, E_, 4, 3, tables)
        d_out = G.make_grad(283 + step, tables, B, D)
        out = a(t(idx), t(off))
        rowidx, tableidx = O.rowidx_from_offsets(off, tables)
        assert_close(out.detach().cpu().numpy(), O.tt_forward(g, B, D, idx, rowidx, tableidx, ref), f"step {step} forward")
        out.backward(t(d_out))
        if adagrad:
            O.tt_backward(g, O.OPTIM_ADAGRAD, B, D, lr, eps, idx, rowidx, tableidx, d_out, ref, state)
        else:
            O.tt_backward(g, O.OPTIM_SGD, B, D, lr, 0.0, idx, rowidx, tableidx, d_out, ref)
        for k in range(3):
            x, y = a.tt_cores[k].detach().cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
            # (the tolerance of test_several_training_steps_track_the_oracle: error compounds over the steps)
            tol = 4 * (2e-6 * np.abs(y).max() + 1e-5 * np.abs(y)) * (8 if adagrad else 1)
            assert (np.abs(x - y) <= tol).all(), f"step {step} core {k}: max err {np.abs(x - y).max():.3e}"
            if adagrad:
                assert_close(a.optimizer_state[k].cpu().numpy(), state[k], f"step {step} state {k}", rtol=4e-5, atol_scale=8e-6)

    one(0)
    one(1)
    a.tt_cores[0].data.mul_(0.5)  # through .data: no version bump
    ref[0] *= np.float32(0.5)
    if adagrad:
        a.optimizer_state[0].data.add_(0.25)
        state[0] += np.float32(0.25)
    one(2)
    one(3)
    sd = {k_: v.clone() for k_, v in a.state_dict().items()}
    with torch.no_grad():
        a.tt_cores[0].zero_()  # (what load_state_dict must undo)
    a.load_state_dict(sd)
    one(4)


def test_rebinding_parameters_and_buffers_reaches_the_native_node(node):
    """Round 3 advisor finding: the module caches the argument lists it hands the C++ node (Parameter / buffer OBJECTS).  Anything
    that re-binds one -- load_state_dict(assign=True), tt_cores[i] = nn.Parameter(..) -- must be seen by the next call: forward
    over the new tensors, the fused optimizer on the new tensors, nothing on the orphaned ones."""
    import tt_embeddings_ops as ops

    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B = 20 * 22 * 25, 64, 64
    kw = dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False, weight_dist="uniform", device=DEV)
    m, other = ops.TTEmbeddingBag(E_, D, r, p, q, **kw), ops.TTEmbeddingBag(E_, D, r, p, q, **kw)
    with torch.no_grad():
        for dst, src in zip(other.tt_cores, G.make_cores(77, 1, p, q, r, "signed")):
            dst.copy_(t(src))
    idx, off = (t(a) for a in G.make_bags(78, B, E_, 6, 2, 1))
    grad = t(G.make_grad(79, 1, B, D)[0])
    m(idx, off).backward(grad)  # (first call: the argument lists are cached now)
    old = [c for c in m.tt_cores]
    old_vals = [c.detach().clone() for c in old]
    m.load_state_dict({k: v.clone() for k, v in other.state_dict().items()}, assign=True)
    assert all(a is not b for a, b in zip(old, m.tt_cores)), "assign=True re-binds the parameters"
    want = other(idx, off)
    got = m(idx, off)
    assert torch.equal(got.detach(), want.detach()), "forward must read the re-bound cores"
    got.backward(grad)
    want.backward(grad)
    torch.cuda.synchronize()
    for a, b in zip(m.tt_cores, other.tt_cores):
        assert torch.equal(a.detach(), b.detach()), "the fused optimizer must train the re-bound cores"
    for a, b in zip(old, old_vals):
        assert torch.equal(a.detach(), b), "... and leave the orphaned ones alone"
    # a single core replaced in place
    with torch.no_grad():
        m.tt_cores[1] = torch.nn.Parameter(other.tt_cores[1].detach().clone() * 0.5)
        other.tt_cores[1].mul_(0.5)
    assert torch.equal(m(idx, off).detach(), other(idx, off).detach())


def test_cache_write_back_on_the_hip_path(node):
    """SURVEY section 8(f3), last clause, on the GPU: cache_populate(write_back=lr) (off by default, not in the reference) takes
    SGD steps on the cores toward what the cached rows learnt before the rows are decompressed anew.  The TT rows of the cached
    keys end up closer to the trained cache rows; write_back=0 -- the reference's behaviour -- leaves the cores bit-identical."""
    import tt_embeddings_ops as ops

    p, q, r = [4, 5, 5], [2, 3, 2], [4, 5]
    E_, D, B, Lp = 100, 12, 16, 4
    m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=True, cache_size=20,
                           hashtbl_size=256, weight_dist="uniform", device=DEV)
    with torch.no_grad():
        for c, src in zip(m.tt_cores, G.make_cores(7, 1, p, q, r, "signed")):
            c.copy_(t(src))
    rs = np.random.RandomState(3)
    off = torch.arange(0, B * Lp + 1, Lp, device=DEV)

    def batch():
        return t((rs.zipf(1.3, size=B * Lp) % E_).astype(np.int64))

    for _ in range(3):
        m(batch(), off)
    m.cache_populate()
    for _ in range(4):  # steady state: hits train their cache rows
        m(batch(), off).backward(t((rs.rand(B, D) * 0.1).astype(np.float32)))
    slots = torch.nonzero(m.cache_state >= 0).flatten()
    keys = m.hashtbl[slots]
    target = m.cache_weight.detach()[m.cache_state[slots].long()].clone()

    def tt_rows():
        full = ops.tt_matrix_to_full(m.tt_p_shapes, m.tt_q_shapes, m.tt_ranks, [c.detach().cpu() for c in m.tt_cores], [1, 0, 2, 3])
        return full[keys.cpu()]

    before = float((tt_rows() - target.cpu()).norm())
    assert before > 1e-3, "the cached rows must have moved away from the cores (otherwise the case shows nothing)"
    cores0 = [c.detach().clone() for c in m.tt_cores]
    m.cache_populate()  # the reference's behaviour: cores untouched
    assert all(torch.equal(a, b) for a, b in zip(cores0, m.tt_cores))
    with torch.no_grad():  # (populate reset the cache rows to the TT rows: restore what they had learnt)
        m.cache_weight[m.cache_state[slots].long()] = target
    m.cache_populate(write_back=2.0, write_back_steps=10)
    torch.cuda.synchronize()
    after = float((tt_rows() - target.cpu()).norm())
    assert after < 0.8 * before, (before, after)


def test_streamed_group_with_a_view_on_top_joins_after_the_lookup_node(node):
    """Round 3 advisor finding: MixedTTEmbeddingBag(streams=True) joins a group's stream from a hook on the group's autograd
    node; a q0 > 4 table (part lookups) returns a VIEW of the lookup's result, and a hook on the ViewBackward node fires before
    the lookup's backward has enqueued its kernels.  The hook must sit on the lookup node: after backward() returns, the caller's
    stream has to be ordered after the group streams' fused-optimizer kernels -- checked by reading the cores on the caller's
    stream right away, against the same steps without streams."""
    import tt_embeddings_ops as ops
    import ttx_mixed

    if node == "python":
        pytest.skip("part lookups (q0 > 4) run through the C++ node")
    D, q, r, B = 512, [8, 8, 8], [16, 16], 64
    Es, ps = [9000, 60000], [[20, 22, 25], [40, 40, 40]]
    kw = dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, weight_dist="uniform", device=DEV)
    torch.manual_seed(5)
    ms = ttx_mixed.MixedTTEmbeddingBag(Es, D, r, ps, q, include_last_offset=False, streams=True, fused=False, **kw)
    mp = ttx_mixed.MixedTTEmbeddingBag(Es, D, r, ps, q, include_last_offset=False, streams=False, fused=False, **kw)
    assert len(ms.groups) == 2 and ms._streams is not None
    with torch.no_grad():
        for a, b in zip(ms.groups, mp.groups):
            for dst, src in zip(b.tt_cores, a.tt_cores):
                dst.copy_(src)
    rs = np.random.RandomState(13)
    grads = [t((rs.rand(B, D) * 0.1).astype(np.float32)) for _ in Es]
    for step in range(3):
        idx, off = [], []
        for e in Es:
            lens = rs.randint(1, 8, size=B)
            off.append(t(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)))
            idx.append(t(rs.randint(0, e, size=int(lens.sum())).astype(np.int64)))
        outs = ms(idx, off)
        assert any("View" in type(o.grad_fn).__name__ or "Select" in type(o.grad_fn).__name__ for o in outs)
        torch.autograd.backward(outs, grads)
        snap = [c.detach().clone() for g in ms.groups for c in g.tt_cores]  # read on the caller's stream, no synchronize
        torch.autograd.backward(mp(idx, off), grads)
        want = [c.detach().clone() for g in mp.groups for c in g.tt_cores]
        for a, b in zip(snap, want):
            assert torch.equal(a, b), f"step {step}: the caller's stream read the cores before the group stream's update"


def test_graphed_step_with_static_buffers_tracks_eager_for_20_steps(node):
    """ttx_graph.GraphedStep: the step captured once behind static input buffers, fed a new batch of the same shape per call --
    20 fused-SGD steps leave the cores bit-identical to the same steps run eagerly (round 3 verdict, item 5)."""
    import tt_embeddings_ops as ops
    import ttx_graph

    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B, Lp = 20 * 22 * 25, 64, 64, 5

    def fresh():
        m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False,
                               weight_dist="uniform", device=DEV)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, G.make_cores(91, 1, p, q, r, "signed")):
                dst.copy_(t(src))
        return m

    me, mg = fresh(), fresh()
    rs = np.random.RandomState(92)
    off = torch.arange(0, B * Lp + 1, Lp, device=DEV)
    batches = [(t(rs.randint(0, E_, size=B * Lp).astype(np.int64)), off, t((rs.rand(B, D) * 0.1).astype(np.float32))) for _ in range(20)]
    before = [c.detach().clone() for c in mg.tt_cores]
    step = ttx_graph.GraphedStep(lambda i, o, g: mg(i, o).backward(g), batches[0], warmup=2)
    with torch.no_grad():  # (warm-up and capture trained on the example batch: back to the common starting point)
        for c, b in zip(mg.tt_cores, before):
            c.copy_(b)
    for i, o, g in batches:
        step(i, o, g)
        me(i, o).backward(g)
    torch.cuda.synchronize()
    for a, b in zip(me.tt_cores, mg.tt_cores):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        step(batches[0][0][:-1], off, batches[0][2])


@pytest.mark.parametrize("optimizer", ["SGD", "EXACT_ADAGRAD"])
@pytest.mark.parametrize("live", [False, True])
def test_backward_of_the_lookups_own_output_past_the_engine(node, optimizer, live, monkeypatch):
    """Round 5: `tt_emb(indices, offsets).backward(grad)` -- the reference benchmark's loop (tt_embeddings_benchmark.py:94-108) --
    with a fused optimizer calls the lookup's node on the calling thread instead of going through autograd's engine
    (tt_embeddings_ops._backward / _direct_register, csrc/ttx_torch.cpp NodeRef).  Same node, same kernels: four steps leave cores, optimizer
    state and cache rows BIT-identical to the engine's route, for both module classes, cache counting and cache live; and
    everything that is not the plain case takes the engine: hooks, retain_graph, dense gradients, weights that need a gradient,
    the output as an operand of further ops."""
    import gc
    import weakref
    import tt_embeddings_ops as ops

    if node != "native":
        pytest.skip("the direct backward belongs to the C++ node")
    # round 6: opt-in.  Importing the module leaves torch.Tensor.backward alone (tests/test_module_cpu.py); this test enables it
    # and puts the original method back whatever happens.
    original = torch.Tensor.backward
    assert original is not ops._backward and not ops.direct_backward_enabled()
    ops.enable_direct_backward()
    try:
        _direct_backward_cases(ops, optimizer, live, monkeypatch, gc, weakref)
    finally:
        ops.disable_direct_backward()
    assert torch.Tensor.backward is original and not ops._direct


def _direct_backward_cases(ops, optimizer, live, monkeypatch, gc, weakref):
    p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
    E_, D, B = 20 * 22 * 25, 64, 64

    def fresh(cls=ops.TTEmbeddingBag, sparse=True):
        # (live: a cache that holds every key of the warm-up batches -- nothing is evicted, so no cached key sits behind an emptied
        #  slot where its re-insert could race a new key's insert for the hit / miss decision, as it can in the reference)
        extra = dict(use_cache=True, cache_size=4096, hashtbl_size=1 << 14) if live else dict(use_cache=False)
        args = (E_, D, r, p, q) if cls is ops.TTEmbeddingBag else (1, E_, D, r, p, q)
        m = cls(*args, sparse=sparse, optimizer=getattr(ops.OptimType, optimizer), learning_rate=0.05, eps=1e-3,
                weight_dist="uniform", device=DEV, **extra)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, G.make_cores(95, 1, p, q, [1] + r + [1], "signed")):
                dst.copy_(t(src))
        return m

    batches = []
    for step in range(4):
        idx, off = G.make_bags(500 + step, B, E_, 6, 3, 1)
        batches.append((t(idx), t(off), t(G.make_grad(600 + step, 1, B, D)[0])))

    def pair(cls=ops.TTEmbeddingBag):
        a, b = fresh(cls), fresh(cls)
        if live:  # count two of the four batches, populate (the other two bring keys the cache does not hold: TT lookups);
            with torch.no_grad():  # b takes a's table and cache as they are
                for i, o, _ in batches[:2]:
                    a(i, o)
            a.cache_populate()
            b.load_state_dict(a.state_dict())
            b.warmup = False
            assert not a.warmup
        return a, b

    def train(m, single):
        taken = []
        for i, o, g in batches:
            out = m(i, o)
            taken.append(id(out) in ops._direct)
            out.backward(g if single else g.unsqueeze(0))
        torch.cuda.synchronize()
        return taken

    for cls in (ops.TTEmbeddingBag, ops.TableBatchedTTEmbeddingBag):
        single = cls is ops.TTEmbeddingBag
        a, b = pair(cls)
        assert all(train(a, single)), "the module is expected to install the direct backward on this route"
        monkeypatch.setattr(ops, "_DIRECT_BACKWARD", False)
        assert not any(train(b, single))
        monkeypatch.setattr(ops, "_DIRECT_BACKWARD", True)
        for x, y in zip(a.tt_cores, b.tt_cores):
            assert torch.equal(x, y)
        for x, y in zip(a.optimizer_state, b.optimizer_state):
            assert torch.equal(x, y)
        if live:  # (cache rows take their updates through float atomics: equal to rounding)
            assert torch.allclose(a.cache_weight, b.cache_weight, rtol=0, atol=1e-6)

    # what is not the plain case goes through the engine, with the engine's semantics
    m, e = pair()
    i, o, g = batches[0]
    seen = []
    out = m(i, o)
    out.register_hook(lambda gr: seen.append(gr.clone()))
    out.backward(g)
    assert len(seen) == 1 and torch.equal(seen[0], g), "a hook on the output must see the gradient"
    monkeypatch.setattr(ops, "_DIRECT_BACKWARD", False)
    e(i, o).backward(g)
    monkeypatch.setattr(ops, "_DIRECT_BACKWARD", True)
    for x, y in zip(m.tt_cores, e.tt_cores):
        assert torch.equal(x, y)
    out = m(i, o)
    out.backward(g, retain_graph=True)  # (the engine's route; the call form must keep working)
    with pytest.raises(RuntimeError):
        m(i, o).backward()  # no gradient for a non-scalar: autograd's own error
    with pytest.raises(RuntimeError):
        m(i, o).backward(g.unsqueeze(0))  # a gradient of another shape than the output: autograd's own error
    (m(i, o) * 2.0).sum().backward()  # the output as an operand
    # an in-place op on the registered output keeps the tensor (and its registry entry) but rebases its grad_fn: the node must see
    # the in-place op's gradient (2 g), i.e. the engine's route -- for the batched module's output and the squeezed view alike
    for cls in (ops.TTEmbeddingBag, ops.TableBatchedTTEmbeddingBag):
        single = cls is ops.TTEmbeddingBag
        a, b = pair(cls)
        gg = g if single else g.unsqueeze(0)
        out = a(i, o)
        assert id(out) in ops._direct
        out.mul_(2.0)
        out.backward(gg)
        monkeypatch.setattr(ops, "_DIRECT_BACKWARD", False)
        ref = b(i, o)
        ref.mul_(2.0)
        ref.backward(gg)
        monkeypatch.setattr(ops, "_DIRECT_BACKWARD", True)
        for x, y in zip(a.tt_cores, b.tt_cores):
            assert torch.equal(x, y), "in-place op on the output before backward(): the direct route must decline"
    # a TorchFunctionMode sees Tensor.backward of a registered output as of any tensor
    from torch.overrides import TorchFunctionMode

    class Spy(TorchFunctionMode):
        calls = 0

        def __torch_function__(self, func, types, args=(), kwargs=None):
            Spy.calls += "backward" in getattr(func, "__name__", "")
            return func(*args, **(kwargs or {}))

    out = m(i, o)
    with Spy():
        out.backward(g)
    assert Spy.calls >= 1, "a torch-function mode must see backward() of a registered output"
    if not live:
        dense = fresh(sparse=False)
        out = dense(i, o)
        assert id(out) not in ops._direct
        out.backward(g)
        assert all(c.grad is not None for c in dense.tt_cores)
        w = torch.rand(i.numel(), device=DEV, requires_grad=True)
        out = m(i, o, per_sample_weights=w)
        assert id(out) not in ops._direct
        out.backward(g)
        assert w.grad is not None
    # nothing is stored ON the tensor (it pickles / saves as any tensor), and its registry entry goes when it goes
    import io
    import pickle
    gc.disable()
    try:
        out = m(i, o)
        assert id(out) in ops._direct and not out.__dict__
        buf = io.BytesIO()
        torch.save(out, buf)
        buf.seek(0)
        assert torch.equal(torch.load(buf, weights_only=False), out)
        assert torch.equal(pickle.loads(pickle.dumps(out)), out)
        key, wr = id(out), weakref.ref(out)
        del out
        assert wr() is None and key not in ops._direct
    finally:
        gc.enable()


_ROUTES = [
    # name, module kwargs (beyond the common ones), p, q, ranks, tables
    ("three cores, r = 32 templates", dict(), [20, 22, 25], [4, 4, 4], [32, 32], 1),
    ("three cores, Adagrad at r = 64", dict(optimizer="EXACT_ADAGRAD"), [20, 22, 25], [4, 4, 8], [64, 64], 1),
    ("generic kernels (ranks 20 / 12)", dict(), [20, 22, 25], [4, 4, 4], [20, 12], 1),
    ("padded factors q = [4, 6, 4]", dict(), [20, 22, 25], [4, 6, 4], [16, 16], 1),
    ("first factor 8 as part lookups", dict(), [6, 7, 8], [8, 4, 4], [16, 16], 1),
    ("three tables in one launch set", dict(), [20, 22, 25], [4, 4, 4], [16, 16], 3),
    ("two cores", dict(), [90, 80], [16, 8], [32], 1),
    ("four cores (MFMA gradient helper)", dict(), [8, 9, 10, 11], [4, 4, 4, 4], [32, 32, 32], 1),
    ("four cores (VALU helper)", dict(), [8, 9, 10, 11], [3, 4, 2, 3], [13, 12, 7], 1),
    ("duplicate lookups share their contraction", dict(dedup=True), [20, 22, 25], [4, 4, 4], [32, 32], 1),
    ("frequency table counting along", dict(use_cache=True, cache_size=256, hashtbl_size=1 << 14), [20, 22, 25], [4, 4, 4], [32, 32], 1),
    # (a cache that holds every key of the three warm-up batches: nothing evicted, so no cached key behind an emptied slot whose
    #  re-insert could race a new key's insert -- the one decision the reference, too, leaves to hardware order)
    ("cache live (hits and misses)", dict(use_cache=True, cache_size=8192, hashtbl_size=1 << 15, live=3), [20, 22, 25], [4, 4, 4], [32, 32], 1),
    # (round 6) ... with the atomic-free cache-row update: the cache rows are BIT-identical too, for SGD and row-wise Adagrad
    ("cache live, sorted cache-row update", dict(use_cache=True, cache_size=8192, hashtbl_size=1 << 15, live=3,
                                                 deterministic_cache_update=True), [20, 22, 25], [4, 4, 4], [32, 32], 1),
    ("cache live, sorted cache-row update, Adagrad", dict(use_cache=True, cache_size=8192, hashtbl_size=1 << 15, live=3, optimizer="EXACT_ADAGRAD",
                                                          deterministic_cache_update=True), [20, 22, 25], [4, 4, 4], [32, 32], 1),
]


@pytest.mark.parametrize("route", _ROUTES, ids=[r[0] for r in _ROUTES])
def test_free_running_training_is_run_to_run_identical(node, route):
    """Round 5 (after a cache-live look-up turned out to be hit or miss by timing, found only because two modules in one state
    were run free and compared): every route, eight fused-optimizer steps on a skewed stream with nothing between them but the
    next call -- twice from the same state: cores and optimizer state BIT-identical.  No float atomics on the TT path, no
    decision taken by timing."""
    import tt_embeddings_ops as ops

    name, extra, p, q, ranks, tables = route
    extra = dict(extra)
    opt = getattr(ops.OptimType, extra.pop("optimizer", "SGD"))
    live = extra.pop("live", 0)
    E_, D, B = int(np.prod(p)), int(np.prod(q)), 192
    cores = G.make_cores(97, tables, p, q, [1] + ranks + [1], "signed")
    rs = np.random.RandomState(98)
    batches = []
    for step in range(8):
        idx, off = G.make_bags(700 + step, B, E_, 8, 4, tables)
        idx = np.where(rs.rand(idx.size) < 0.3, idx[rs.randint(0, idx.size, idx.size)] % 50, idx)  # (a hot head: duplicates, hot slices)
        g = G.make_grad(800 + step, tables, B, D)
        batches.append((t(idx), t(off), t(g if tables > 1 else g[0])))

    def run():
        kw = dict(sparse=True, optimizer=opt, learning_rate=0.05, eps=1e-3, weight_dist="uniform", device=DEV, use_cache=False)
        kw.update(extra)
        if tables > 1:
            m = ops.TableBatchedTTEmbeddingBag(tables, E_, D, ranks, p, q, **kw)
        else:
            m = ops.TTEmbeddingBag(E_, D, ranks, p, q, **kw)
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, cores):
                dst.copy_(t(src))
        if live:
            with torch.no_grad():
                for i, o, _ in batches[:live]:
                    m(i, o)
            m.cache_populate()
            assert not m.warmup
        for i, o, g in batches:
            m(i, o).backward(g)
        torch.cuda.synchronize()
        rows = None
        if live:  # the cached rows BY KEY (which cache row a key got depends on where racing inserts seated the keys of equal count)
            k_, s_, w_ = m.hashtbl.cpu().numpy(), m.cache_state.cpu().numpy(), m.cache_weight.detach().cpu().numpy()
            sel = (k_ >= 0) & (s_ >= 0)
            order = np.argsort(k_[sel])
            rows = (k_[sel][order], w_[s_[sel][order]])
        return [c.detach().clone() for c in m.tt_cores] + [s.detach().clone() for s in m.optimizer_state], rows

    (a, ca), (b, cb) = run(), run()
    for k, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"{name}: tensor {k} differs between two free-running runs (max {float((x - y).abs().max()):.3e})"
    if live:  # (cache rows take their updates through float atomics: equal to rounding -- a hot row sums ~500 terms per step in
        #  hardware order; a hit / miss flip would be ~5e-3)
        assert np.array_equal(ca[0], cb[0]), "the same keys are cached"
        err = np.abs(ca[1] - cb[1]).max()
        if extra.get("deterministic_cache_update"):
            assert np.array_equal(ca[1], cb[1]), f"sorted cache-row update: cache rows by key differ between two runs (max {err:.3e})"
        assert err <= 2e-5, f"cache rows by key: max difference {err:.3e}"
