"""Run the REFERENCE's own property tests (/root/reference/tt_embeddings_test.py:62-525) on top of the CPU oracle.

TEST INFRASTRUCTURE, BUILD CONTAINER ONLY: needs /root/reference (absent on the GPU box) and is started as a process of its own by
tests/test_reference_own_tests.py (the reference's Python module is called `tt_embeddings_ops` like the product's; the two must not
meet in one interpreter).  Nothing of the reference is copied: its test file and its tt_embeddings_ops.py are imported from where
they lie.  What is swapped in:
  * `tt_embeddings` (the reference's native CUDA extension, unbuildable here)  ->  tests/oracle_engine.py, the 11-function surface
    backed by oracle/ttx_oracle.c;
  * "cuda:0" -> the CPU: torch.cuda.is_available() says yes (the reference asserts it, tt_embeddings_ops.py:454), current_device()
    is the CPU device (ops.py:526-595 allocate with it), torch.device("cuda:0") resolves to the CPU, set_device() does nothing.
So the six hypothesis tests -- forward, dense backward, fused SGD, fused Adagrad, table-batched forward and backward, each against
torch.nn.EmbeddingBag on tt_emb.full_weight() with autograd through tt_matrix_to_full -- pin (a) the oracle's arithmetic and (b) the
call order and argument meaning of the 11 functions as the reference's own module code drives them (SURVEY.md section 8 rows a2 / a14 / c).
Exit code 0 = all six passed.  `--examples N` caps hypothesis' examples per test (the reference asks for 20)."""
import argparse
import os
import sys
import types
import unittest

REF = os.environ.get("TTX_REFERENCE_DIR", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def describe(x):
    import torch

    if isinstance(x, torch.Tensor):
        return ["T", str(x.dtype).replace("torch.", ""), list(x.shape)]
    if isinstance(x, (list, tuple, torch.nn.ParameterList)):  # (the reference hands cache_populate its ParameterList)
        return [describe(v) for v in x]
    if isinstance(x, (bool, int, float, str)) or x is None:
        return x
    return str(x)


def life_cycle(ops, engine_module, torch, extra_kwargs=None):
    """One cache life cycle on a TTEmbeddingBag of `ops` (the reference's module, or -- called from the pytest side -- the
    product's on the same stand-in engine): two counting steps, populate, two cache-live steps, for the fused SGD and the dense
    mode.  -> (trace of native-module calls, list of result arrays)."""
    import numpy as np

    trace = []
    for name in ("tt_forward", "tt_dense_backward", "tt_sgd_backward", "tt_adagrad_backward", "update_cache_state", "cache_populate",
                 "preprocess_indices_sync", "cache_forward", "cache_backward_sgd", "cache_backward_dense",
                 "cache_backward_rowwise_adagrad_approx"):
        fn = getattr(engine_module, name)

        def traced(*args, _fn=fn, _name=name, **kwargs):
            trace.append([_name, [describe(v) for v in args], sorted(kwargs)])
            return _fn(*args, **kwargs)
        setattr(engine_module, name, traced)
    p, q, r = [7, 9, 11], [3, 4, 5], [13, 12]
    E_, D, B = 7 * 9 * 11, 60, 24
    results = []
    for mode in ("sgd", "adagrad", "dense"):
        trace.append(["-- mode", mode])
        rs = np.random.RandomState(7)
        kw = dict(sparse=mode != "dense", optimizer=ops.OptimType.EXACT_ADAGRAD if mode == "adagrad" else ops.OptimType.SGD,
                  learning_rate=0.05, eps=1e-3, use_cache=True, cache_size=64, hashtbl_size=1024, weight_dist="uniform")
        kw.update(extra_kwargs or {})
        torch.manual_seed(3)
        m = ops.TTEmbeddingBag(E_, D, r, p, q, **kw)
        cores = [rs.uniform(-0.5, 0.5, size=tuple(c.shape)).astype(np.float32) for c in m.tt_cores]
        with torch.no_grad():
            for dst, src in zip(m.tt_cores, cores):
                dst.copy_(torch.from_numpy(src))
        for step in range(4):
            if step == 2:
                trace.append(["-- cache_populate()"])
                m.cache_populate()
            lengths = rs.randint(0, 6, size=B)
            idx = torch.from_numpy((rs.zipf(1.4, size=int(lengths.sum())) % E_).astype(np.int64))
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64))
            trace.append(["-- step", step])
            out = m(idx, off)
            out.backward(torch.from_numpy(rs.uniform(-0.1, 0.1, size=(B, D)).astype(np.float32)))
            results.append(out.detach().numpy().copy())
        results += [c.detach().numpy().copy() for c in m.tt_cores]
        results.append(m.cache_weight.detach().numpy().copy())
        if mode == "dense":
            results += [c.grad.numpy().copy() for c in m.tt_cores]
    return trace, results


def trace_life_cycle(ref_ops, stand_in, torch) -> int:
    import json

    import numpy as np

    trace, results = life_cycle(ref_ops, stand_in, torch)
    out = os.environ.get("TTX_TRACE_NPZ")
    if out:
        np.savez(out, *results)
    print(json.dumps(trace))
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--examples", type=int, default=20)
    ap.add_argument("-k", default=None, help="run only the tests whose name contains this")
    ap.add_argument("--trace", action="store_true",
                    help="instead of the tests: drive the reference's TTEmbeddingBag through a cache life cycle and print, as one JSON "
                         "line, the sequence of native-module calls it makes (names, scalar arguments, tensor shapes) and its results")
    a = ap.parse_args()
    if not os.path.exists(os.path.join(REF, "tt_embeddings_test.py")):
        print("reference absent: nothing to run")
        return 77
    sys.path.insert(0, HERE)  # oracle_engine / oracle_lib
    import torch

    import oracle_engine

    cpu = torch.device("cpu")
    real_device = torch.device

    class _DeviceMeta(type):
        def __instancecheck__(cls, obj):
            return isinstance(obj, real_device)

    class _Device(metaclass=_DeviceMeta):
        def __new__(cls, *args, **kwargs):
            d = real_device(*args, **kwargs)
            return cpu if d.type == "cuda" else d

    torch.device = _Device
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *_a, **_k: None
    torch.cuda.current_device = lambda: cpu
    torch.cuda.synchronize = lambda *_a, **_k: None
    # the reference's native module name -> the oracle-backed stand-in (same 11 names, same argument order)
    stand_in = types.ModuleType("tt_embeddings")
    for name in ("tt_forward", "tt_dense_backward", "tt_sgd_backward", "tt_adagrad_backward", "update_cache_state", "cache_populate",
                 "preprocess_indices_sync", "cache_forward", "cache_backward_sgd", "cache_backward_dense",
                 "cache_backward_rowwise_adagrad_approx"):
        setattr(stand_in, name, getattr(oracle_engine, name))
    sys.modules["tt_embeddings"] = stand_in
    calls = {}
    for name in list(vars(stand_in)):
        fn = getattr(stand_in, name)
        if callable(fn):
            def counted(*args, _fn=fn, _name=name, **kwargs):
                calls[_name] = calls.get(_name, 0) + 1
                return _fn(*args, **kwargs)
            setattr(stand_in, name, counted)
    sys.path.insert(0, REF)  # the reference's tt_embeddings_ops.py and tt_embeddings_test.py, where they lie
    import tt_embeddings_ops as ref_ops

    assert os.path.dirname(os.path.abspath(ref_ops.__file__)) == os.path.abspath(REF), "not the reference's module"
    import tt_embeddings_test as ref_tests
    from hypothesis import settings

    if a.trace:
        return trace_life_cycle(ref_ops, stand_in, torch)
    case = ref_tests.TestTTEmbeddingBag
    names = [n for n in unittest.defaultTestLoader.getTestCaseNames(case) if a.k is None or a.k in n]
    if a.examples != 20:  # (the @settings object hypothesis keeps on the wrapped test function: the same with fewer examples)
        for n in names:
            fn = getattr(case, n)
            fn._hypothesis_internal_use_settings = settings(parent=fn._hypothesis_internal_use_settings, max_examples=a.examples)
    suite = unittest.TestSuite(case(n) for n in names)
    res = unittest.TextTestRunner(verbosity=1, stream=sys.stderr).run(suite)
    print(f"reference tests run: {res.testsRun}, failures: {len(res.failures)}, errors: {len(res.errors)}, skipped: {len(res.skipped)}")
    print("native-module calls: " + ", ".join(f"{k}={v}" for k, v in sorted(calls.items())))
    return 0 if res.wasSuccessful() and res.testsRun == len(names) and not res.skipped else 1


if __name__ == "__main__":
    sys.exit(main())
