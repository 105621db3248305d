"""Fixed-seed slices of the randomised parity sweeps (tests/fuzz_cases.py) as part of the GPU suite:
250 geometry/batch cases of the TT path (every plan route must be hit) and 250 of the cache path."""
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_plan_and_contraction_fuzz_slice(seed):
    n, routes = fuzz_cases.run_plan_cases(seed=seed, max_cases=125 if seed < 2 else 60)
    assert n == (125 if seed < 2 else 60)
    must = ("tiny", "single", "units", "wide", "multi-pass", "mixed", "prologue", "block-walk") if seed == 0 else \
        (("tiny", "single", "mixed", "prologue", "block-walk") if seed == 1 else ("tiny", "single"))
    for route in must:
        assert routes.get(route, 0) > 0, f"plan route {route!r} not exercised: {routes}"


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_cache_prologue_gather_scatter_fuzz_slice(seed):
    n = 125 if seed < 2 else 60
    assert fuzz_cases.run_cache_cases(seed=seed, max_cases=n) == n
