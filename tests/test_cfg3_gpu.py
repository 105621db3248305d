"""BASELINE configs[2] at FULL size: cfg2's table with the software cache live -- hashtbl / cache_freq int64[2^20],
cache_state int32[2^20], cache_weight fp32[262144, 64] (64 MiB), Zipf(1.2) lookups.

The hash-table state is built by the CPU oracle from 4.1 M Zipf draws (sequential inserts -> one defined state; the
racing insert order of a GPU update is covered in test_cache_gpu.py / test_refdev_gpu.py), then:
  * cache_populate at full size (multi-block 64-bit radix sort of 2^20 pairs sized by the largest frequency,
    mark / evict, 262,144 rows through ttx_tt_rows): table state bit-exact, rows to 1e-5;
  * one benchmark batch (512 bags x 20) through the cache-live path: lookup + stable partition bit-exact, hit gather,
    contraction of the misses, SGD scatter into the cache rows and fused SGD on the cores, against the oracle;
  * the same step through the module (both host routes)."""
import numpy as np
import pytest
import torch

import gen_inputs as G
import oracle_lib as O
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, CS, E_, D = 1 << 20, 1 << 18, 11_000_000, 64
P, Q, R = G.CFG2["p"], G.CFG2["q"], G.pad_ranks(G.CFG2["ranks"], 3)
LR = 0.1


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def live():
    """hash-table state after the warm-up stream, the oracle's populate result and the GPU's"""
    import tt_embeddings as E

    rs = np.random.RandomState(1234)
    keys, freq = np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)
    for _ in range(100):
        O.update_cache_state((rs.zipf(1.2, size=40960).astype(np.int64)) % E_, keys, freq)
    assert int((keys != -1).sum()) > CS, "the warm-up stream must hold more distinct keys than cache rows"
    cores = G.make_cores(1234, 1, P, Q, R, "uniform")
    state = np.full(H, -1, dtype=np.int32)
    w = np.zeros((CS, D), dtype=np.float32)
    dk, df, ds, dw = t(keys), t(freq), t(state), t(w)
    E.cache_populate(E_, P, Q, R, [t(c) for c in cores], torch.zeros(3, dtype=torch.int64, device=DEV), dk, df, ds, dw)
    O.cache_populate(O.make_geom(1, P, Q, R), cores, keys, freq, state, w)
    return dict(rs=rs, cores=cores, keys=keys, freq=freq, state=state, w=w, dk=dk, df=df, ds=ds, dw=dw)


def test_populate_full_size(live):
    assert np.array_equal(live["dk"].cpu().numpy(), live["keys"]), "hashtbl after eviction"
    assert np.array_equal(live["df"].cpu().numpy(), live["freq"]), "cache_freq after eviction"
    assert np.array_equal(live["ds"].cpu().numpy(), live["state"]), "cache_state (slot -> cache row)"
    assert int((live["state"] >= 0).sum()) == CS
    assert_close(live["dw"].cpu().numpy(), live["w"], "262,144 decompressed cache rows")


def _batch(live):
    rs = np.random.RandomState(99)
    idx = (rs.zipf(1.2, size=10240).astype(np.int64)) % E_
    off = np.arange(0, 10241, 20, dtype=np.int64)
    grad = (rs.rand(512, D) * 0.1).astype(np.float32)
    return idx, off, grad


def _oracle_step(live, idx, off, grad, count_first):
    """the reference's cache-live step (tt_embeddings_ops.py:821-874, :179-356) on the oracle: frequency update,
    lookup + partition, contraction of the TT entries + gather of the hits; backward: fused SGD on the cores from
    the TT entries, SGD scatter into the cache rows from the hits"""
    keys, freq = live["keys"].copy(), live["freq"].copy()
    if count_first:
        O.update_cache_state(idx, keys, freq)
    pc, pr, tb, ntt, loc = O.preprocess_indices(idx, off, 1, False, keys, live["state"])
    g = O.make_geom(1, P, Q, R)
    out = O.tt_forward(g, 512, D, pc, pr, tb, live["cores"], nnz=ntt)
    O.cache_forward(512, loc[ntt:], pr[ntt:], live["w"], out[0])
    cores = [c.copy() for c in live["cores"]]
    O.tt_backward(g, O.OPTIM_SGD, 512, D, LR, 0.0, pc, pr, tb, grad[None], cores, nnz=ntt)
    w64 = live["w"].astype(np.float64)
    np.subtract.at(w64, loc[ntt:], np.float64(np.float32(LR)) * grad[pr[ntt:]].astype(np.float64))
    return dict(keys=keys, freq=freq, pc=pc, pr=pr, ntt=ntt, loc=loc, out=out, cores=cores, w64=w64)


def test_cache_live_step_entry_points(live):
    import tt_embeddings as E

    idx, off, grad = _batch(live)
    exp = _oracle_step(live, idx, off, grad, count_first=False)
    got = E.preprocess_indices_sync(t(idx), t(off), 1, False, live["dk"], live["ds"])
    ntt = exp["ntt"]
    assert got[3] == ntt
    hit = 1.0 - ntt / idx.size
    assert 0.5 < hit < 0.99, f"hit rate {hit:.3f}: expected a mix dominated by hits (Zipf 1.2, 256Ki rows)"
    assert np.array_equal(got[0].cpu().numpy(), exp["pc"]), "partitioned colidx"
    assert np.array_equal(got[1].cpu().numpy(), exp["pr"]), "partitioned rowidx"
    assert np.array_equal(got[4].cpu().numpy()[ntt:], exp["loc"][ntt:]), "cache locations"
    cores = [t(c) for c in live["cores"]]
    Lt = torch.zeros(3, dtype=torch.int64, device=DEV)
    out = E.tt_forward(1000, 1, 512, D, P, Q, R, Lt, ntt, got[0], got[1], got[2], cores)
    E.cache_forward(512, idx.size - ntt, got[4][ntt:], got[1][ntt:], live["dw"], out)
    assert_close(out.cpu().numpy(), exp["out"], "cache-live output (TT rows + cache rows)")
    E.tt_sgd_backward(1000, D, LR, P, Q, R, Lt, ntt, got[0], got[1], got[2], t(grad[None]), cores)
    for k in range(3):
        assert_close(cores[k].cpu().numpy(), exp["cores"][k], f"fused SGD core{k} from the TT entries")
    dw = live["dw"].clone()
    E.cache_backward_sgd(idx.size - ntt, t(grad), got[4][ntt:], got[1][ntt:], LR, dw)
    # (the hottest cache row takes ~1,700 adds of this batch: compared against the float64 sum)
    assert_close(dw.cpu().numpy(), exp["w64"], "SGD scatter into the cache rows")


@pytest.mark.parametrize("route", ["native", "python"])
def test_cache_live_step_through_the_module(live, route, monkeypatch):
    import tt_embeddings_ops as ops

    if route == "python":
        monkeypatch.setenv("TTX_NO_NATIVE_NODE", "1")
    else:
        monkeypatch.delenv("TTX_NO_NATIVE_NODE", raising=False)
        assert ops._native_node() is not None
    idx, off, grad = _batch(live)
    exp = _oracle_step(live, idx, off, grad, count_first=True)
    m = ops.TTEmbeddingBag(E_, D, R[1:-1], P, Q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR, use_cache=True,
                           cache_size=CS, hashtbl_size=H, weight_dist="uniform", device=DEV)
    with torch.no_grad():
        for dst, src in zip(m.tt_cores, live["cores"]):
            dst.copy_(t(src))
        m.hashtbl.copy_(live["dk"])
        m.cache_freq.copy_(live["df"])
        m.cache_state.copy_(live["ds"])
        m.cache_weight.copy_(live["dw"])
    m.warmup = False
    out = m(t(idx), t(off))
    out.backward(t(grad))
    assert_close(out.detach().cpu().numpy(), exp["out"][0], f"module ({route}) cache-live output")
    for k in range(3):
        assert_close(m.tt_cores[k].detach().cpu().numpy(), exp["cores"][k], f"module ({route}) fused SGD core{k}")
    assert_close(m.cache_weight.detach().cpu().numpy(), exp["w64"], f"module ({route}) cache rows after SGD")
    # the step also counted its indices (tt_embeddings_ops.py:827-833): order-free content equals the oracle's
    fk, ff = m.hashtbl.cpu().numpy(), m.cache_freq.cpu().numpy()
    assert int(ff.sum()) == int(exp["freq"].sum()) or abs(int(ff.sum()) - int(exp["freq"].sum())) < 20
    both = (fk == exp["keys"])
    assert both.mean() > 0.9999 and np.array_equal(ff[both], exp["freq"][both])
