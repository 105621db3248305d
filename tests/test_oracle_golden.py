"""CPU tests (no GPU): the oracle against every golden vector the build holds --
the reference's Python oracle on its own test shapes (tests/golden/small_cases.npz),
the benchmark configs (cfg2 / cfg4), and the hash-table known-answer vectors."""
import json
import os

import numpy as np
import pytest

import gen_inputs as G
import oracle_lib as O
from util import LR, EPS, adagrad_expected, assert_adagrad_close, assert_close, sgd_expected

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(c, mode):
    g = O.make_geom(c["tables"], c["p"], c["q"], c["r"])
    rowidx, tableidx = O.rowidx_from_offsets(c["offsets"], c["tables"])
    cores = [x.copy() for x in c["cores"]]
    out = O.tt_forward(g, c["B"], c["D"], c["indices"], rowidx, tableidx, cores)
    res = dict(out=out, cores=cores)
    if mode == "dense":
        res["grads"] = O.tt_backward(g, O.OPTIM_DENSE, c["B"], c["D"], 0, 0, c["indices"], rowidx, tableidx, c["d_out"], cores)
    elif mode == "sgd":
        O.tt_backward(g, O.OPTIM_SGD, c["B"], c["D"], LR, 0, c["indices"], rowidx, tableidx, c["d_out"], cores)
    elif mode == "adagrad":
        res["state"] = [np.zeros_like(x) for x in cores]
        O.tt_backward(g, O.OPTIM_ADAGRAD, c["B"], c["D"], LR, EPS, c["indices"], rowidx, tableidx, c["d_out"], cores, res["state"])
    return res


def test_small_cases_forward_and_dense(small_cases):
    for name, c in small_cases.items():
        r = _run(c, "dense")
        assert_close(r["out"], c["out"], f"{name} out")
        for k in range(c["T"]):
            assert_close(r["grads"][k], c["grads"][k], f"{name} grad{k}")


def test_small_cases_sgd_adagrad(small_cases):
    for name, c in small_cases.items():
        r = _run(c, "sgd")
        for k, e in enumerate(sgd_expected(c["cores"], c["grads"])):
            assert_close(r["cores"][k], e, f"{name} sgd{k}")
        r = _run(c, "adagrad")
        exp, st = adagrad_expected(c["cores"], c["grads"])
        for k in range(c["T"]):
            assert_close(r["state"][k], st[k], f"{name} state{k}")
            assert_adagrad_close(r["cores"][k], exp[k], c["grads"][k], f"{name} ada{k}")


def test_round4_cases(round4_cases):
    """the geometry classes round 4 moved onto new routes (tests/golden/round4_cases.npz, expanded by the reference's Python):
    the oracle first -- the GPU tests compare the new routes with these vectors AND with the oracle"""
    test_small_cases_forward_and_dense(round4_cases)
    test_small_cases_sgd_adagrad(round4_cases)


def big_case(tag):
    cfg, seed = {"cfg2": (G.CFG2, 1234), "cfg4": (G.CFG4, 4321), "r128": (G.R128, 2468)}[tag]
    z = np.load(os.path.join(HERE, "golden", f"{tag}.npz"))
    assert int(z["seed"][0]) == seed
    p, q, r = cfg["p"], cfg["q"], G.pad_ranks(cfg["ranks"], 3)
    E_, D = int(np.prod(p)), int(np.prod(q))
    idx, off = G.make_requests(seed + 1, 1, cfg["B"], 1, cfg["L"], E_)[0]
    c = dict(tables=1, T=3, p=p, q=q, r=r, B=cfg["B"], D=D, indices=idx, offsets=off,
             cores=G.make_cores(seed, 1, p, q, r, "uniform"), d_out=G.make_grad(seed + 2, 1, cfg["B"], D))
    return c, z


def check_big(tag, res_dense, res_sgd=None, res_ada=None):
    """compare a full-size run with the golden sub-samples and checksums"""
    c, z = big_case(tag)
    assert_close(res_dense["out"], z["out"], f"{tag} out")
    for k in range(3):
        g = res_dense["grads"][k].reshape(-1, res_dense["grads"][k].shape[-1])
        rows = z[f"grad{k}_rows"]
        assert_close(g[rows], z[f"grad{k}_sub"], f"{tag} grad{k} rows")
        assert_close(g.astype(np.float64).sum(axis=1), z[f"grad{k}_rowsum"], f"{tag} grad{k} per-slice sums", rtol=2e-5)
        s = z[f"grad{k}_sum"]
        assert abs(g.astype(np.float64).sum() - s[0]) <= 1e-5 * abs(s[0])
        assert abs((g.astype(np.float64) ** 2).sum() - s[1]) <= 2e-5 * abs(s[1])
        if res_sgd is not None:
            assert_close(res_sgd["cores"][k].reshape(g.shape)[rows], z[f"sgd{k}_sub"], f"{tag} sgd{k} rows")
        if res_ada is not None:
            assert_adagrad_close(res_ada["cores"][k].reshape(g.shape)[rows], z[f"ada{k}_sub"], z[f"grad{k}_sub"], f"{tag} ada{k} rows")


@pytest.mark.parametrize("tag", ["cfg2", "cfg4", "r128"])
def test_benchmark_configs(tag):
    c, _ = big_case(tag)
    check_big(tag, _run(c, "dense"), _run(c, "sgd") if tag != "cfg4" else None, _run(c, "adagrad") if tag != "cfg2" else None)


def test_hash_known_answers():
    kat = json.load(open(os.path.join(HERE, "golden", "hashtbl_kat.json")))
    for key, slots in kat["hash64"].items():
        assert [O.hash64(int(key), s) for s in kat["sizes"]] == slots, key
    for key, raw in kat["hash64_raw"].items():
        assert O.hash64_raw(int(key)) == int(raw, 16)
    assert O.hash32(12345, 1000) == kat["hash32_12345_1000"]
    seq = kat["insert"]
    keys = np.full(seq["size"], -1, dtype=np.int64)
    vals = np.zeros(seq["size"], dtype=np.int64)
    assert [O.hashtbl_insert(k, 1, keys, vals) for k in seq["keys"]] == seq["returns"]
    assert keys.tolist() == seq["final_keys"] and vals.tolist() == seq["final_freqs"]
    for k, want in seq["find"].items():
        assert O.hashtbl_find(int(k), keys) == want


def test_hash_vs_reference_build():
    """oracle/_ref/libhashref.so = the reference's own murmor_hash_3_32 compiled
    from /root/reference/hashtbl_cuda_utils.cuh (oracle/Makefile)."""
    import ctypes as C

    so = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libhashref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = C.CDLL(so)
    ref.ref_hash64.restype = C.c_uint32
    ref.ref_hash64.argtypes = [C.c_int64, C.c_int32]
    ref.ref_hash32.restype = C.c_uint32
    ref.ref_hash32.argtypes = [C.c_int32, C.c_int32]
    rs = np.random.RandomState(3)
    keys = np.concatenate([rs.randint(-2 ** 62, 2 ** 62, size=4000), np.arange(-5, 200), [2 ** 63 - 1, -2 ** 63]])
    for size in (1, 16, 1000, 1 << 20, 11_000_000, 2 ** 31 - 1):
        for k in keys:
            assert O.hash64(int(k), size) == ref.ref_hash64(int(k), size)
    for k in rs.randint(-2 ** 31, 2 ** 31, size=2000):
        assert O.hash32(int(k), 1000003) == ref.ref_hash32(int(k), 1000003)


def check_cfg5_tables(c, out, grads=None, sgd_cores=None):
    """a 26-table run at B = 512 (out [26, 512, 64], dense core gradients / SGD-updated cores [26, p_t, slice]) against
    the reference's results for the tables stored in tests/golden/cfg5.npz"""
    z = np.load(os.path.join(HERE, "golden", "cfg5.npz"))
    assert int(z["seed"][0]) == G.CFG5_SEED and c["B"] == 512
    for k in z["tables"].tolist():
        assert_close(out[k], z[f"t{k}_out"], f"cfg5 table {k} out")
        for t in range(3):
            rows = z[f"t{k}_grad{t}_rows"]
            if grads is not None:
                assert_close(grads[t][k][rows], z[f"t{k}_grad{t}_sub"], f"cfg5 table {k} grad{t} rows")
                assert_close(grads[t][k].astype(np.float64).sum(axis=1), z[f"t{k}_grad{t}_rowsum"], f"cfg5 table {k} grad{t} per-slice sums",
                             rtol=2e-5)
            if sgd_cores is not None:
                assert_close(sgd_cores[t][k][rows], z[f"t{k}_sgd{t}_sub"], f"cfg5 table {k} sgd{t} rows")


def test_cfg5_tables_vs_reference_golden():
    """BASELINE configs[4] geometry: the oracle on the three golden tables of the 26 (one table at a time: the tables of a
    batched lookup are independent)"""
    c = G.cfg5_case(512)
    out = np.zeros((c["tables"], 512, c["D"]), dtype=np.float32)
    grads = [np.zeros_like(x) for x in c["cores"]]
    sgd = [x.copy() for x in c["cores"]]
    for k in G.CFG5_GOLDEN_TABLES:
        ck = G.table_of(c, k)
        ck["cores"] = [np.ascontiguousarray(x) for x in ck["cores"]]
        r = _run(ck, "dense")
        out[k] = r["out"][0]
        for t in range(3):
            grads[t][k] = r["grads"][t][0]
            sgd[t][k] = _run(ck, "sgd")["cores"][t][0]
    check_cfg5_tables(c, out, grads, sgd)


def test_all_cores_baseline_equals_the_sequential_oracle(small_cases):
    """oracle/ttx_cpu_baseline.c (OpenMP; what bench.py times as cpu_baseline) against the sequential oracle:
    forward bit-identical (same per-bag order), fused SGD / Adagrad to rounding (thread-partial gradient sums)"""
    for name in ("t3_tb3_s1", "t2_tb1_s0", "t4_tb3_s0"):
        c = small_cases[name]
        g = O.make_geom(c["tables"], c["p"], c["q"], c["r"])
        rowidx, tableidx = O.rowidx_from_offsets(c["offsets"], c["tables"])
        for optim in (O.OPTIM_SGD, O.OPTIM_ADAGRAD):
            ref_cores = [x.copy() for x in c["cores"]]
            ref_state = [np.zeros_like(x) for x in ref_cores]
            ref_out = O.tt_forward(g, c["B"], c["D"], c["indices"], rowidx, tableidx, ref_cores)
            O.tt_backward(g, optim, c["B"], c["D"], LR, EPS, c["indices"], rowidx, tableidx, c["d_out"], ref_cores, ref_state)
            cores = [x.copy() for x in c["cores"]]
            state = [np.zeros_like(x) for x in cores]
            step = O.OmpStep(g, c["B"], c["D"], c["indices"].size, cores)
            out = step(optim, LR, EPS, c["indices"], c["offsets"], rowidx, tableidx, c["d_out"], cores, state)
            assert np.array_equal(out, ref_out), f"{name}: forward of the all-cores build"
            for k in range(c["T"]):
                if optim == O.OPTIM_SGD:
                    assert_close(cores[k], ref_cores[k], f"{name} omp sgd core{k}")
                else:
                    assert_close(state[k], ref_state[k], f"{name} omp adagrad state{k}")


def test_oracle_rowwise_adagrad_vs_float64_restatement_of_the_reference_kernel():
    """a12 (round-5 verdict): oracle/ttx_oracle.c's cache_backward_rowwise_adagrad_approx against an INDEPENDENT float64
    restatement of tt_embeddings_cuda.cu:1735-1795 written from the kernel's segment structure (util.rowwise_adagrad_segments_f64):
    a stream where rows are hit from several bags (the oracle's sequential order = the restatement's segment order) and one where
    the reference itself has a single possible result (no row in two bags)."""
    import oracle_lib as O
    from util import rowwise_adagrad_segments_f64

    rs = np.random.RandomState(11)
    for cs, B, D, n, zipf in ((200, 64, 64, 1500, True), (4000, 128, 60, 900, False)):
        if zipf:
            loc = ((rs.zipf(1.2, size=n) - 1) % cs).astype(np.int32)
            rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
        else:
            rowidx = np.sort(rs.randint(0, B, size=n)).astype(np.int64)
            loc = (rowidx * 31 + rs.randint(0, 31, size=n)).astype(np.int32)  # bag b owns rows [31 b, 31 b + 31)
        w = rs.randn(cs, D).astype(np.float32)
        grad = ((rs.rand(B, D) - 0.5) * 0.2).astype(np.float32)
        st0 = (rs.rand(cs) * 0.01).astype(np.float32)
        st64, w64 = rowwise_adagrad_segments_f64(grad, loc, rowidx, 0.1, 1e-4, st0, w)
        st_o, w_o = st0.copy(), w.copy()
        O.cache_backward_rowwise_adagrad_approx(grad, loc, rowidx, 0.1, 1e-4, st_o, w_o)
        assert_close(st_o, st64, "rowwise adagrad state", rtol=2e-5, atol_scale=4e-6)
        assert_close(w_o, w64, "rowwise adagrad weights", rtol=2e-5, atol_scale=4e-6)
