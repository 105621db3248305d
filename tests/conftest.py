import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "fbtt-embedding_amd")
for p in (HERE, PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load_cases(fname):
    import numpy as np

    z = np.load(os.path.join(HERE, "golden", fname))
    names = sorted({k.split("/")[0] for k in z.files})
    cases = {}
    for name in names:
        meta = z[f"{name}/meta"]
        tables, T = int(meta[0]), int(meta[1])
        c = dict(
            tables=tables, T=T,
            p=meta[2:2 + T].tolist(), q=meta[2 + T:2 + 2 * T].tolist(), r=meta[2 + 2 * T:].tolist(),
            indices=z[f"{name}/indices"], offsets=z[f"{name}/offsets"], d_out=z[f"{name}/d_out"], out=z[f"{name}/out"],
            cores=[z[f"{name}/core{t}"] for t in range(T)], grads=[z[f"{name}/grad{t}"] for t in range(T)],
        )
        if len(c["r"]) == T - 1:  # stored unpadded -> [1, r1, .., 1]
            c["r"] = [1] + c["r"] + [1]
        c["B"] = (c["offsets"].size - 1) // tables
        c["D"] = int(np.prod(c["q"]))
        cases[name] = c
    return cases


@pytest.fixture(scope="session")
def small_cases():
    """the reference's own test shapes (tt_embeddings_test.py:65-70) + the README toy: tests/golden/small_cases.npz"""
    return _load_cases("small_cases.npz")


@pytest.fixture(scope="session")
def round4_cases():
    """the geometry classes round 4 moved onto new routes, expanded by the reference's Python (tests/golden/make_golden.py
    round4_cases): tests/golden/round4_cases.npz"""
    return _load_cases("round4_cases.npz")
