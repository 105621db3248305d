"""helpers shared by the parity tests"""
import numpy as np

# fp32 tolerance of BASELINE.json's north_star: 1e-5 rtol.  atol is tied to the
# magnitude of the reference tensor (sums of a few hundred fp32 products).
RTOL = 1e-5
ATOL_SCALE = 2e-6

LR, EPS = 0.1, 1.0e-4  # tt_embeddings_test.py:268-269


def assert_close(got, ref, what="", rtol=RTOL, atol_scale=ATOL_SCALE):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(ref).all(), f"{what}: reference has non-finite values"
    assert np.isfinite(got).all(), f"{what}: result has non-finite values"
    atol = atol_scale * max(float(np.abs(ref).max()) if ref.size else 0.0, 1e-30)
    err = np.abs(got - ref)
    bad = err > atol + rtol * np.abs(ref)
    if bad.any():
        i = np.unravel_index(np.argmax(err - (atol + rtol * np.abs(ref))), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.size} elements out of tolerance (rtol {rtol}, atol {atol:.3e}); "
            f"worst at {i}: got {got[i]!r} ref {ref[i]!r}")


def sgd_expected(cores, grads, lr=LR):
    """tt_embeddings_test.py:243-246 in fp32"""
    return [c - g * np.float32(lr) for c, g in zip(cores, grads)]


def adagrad_expected(cores, grads, lr=LR, eps=EPS):
    """tt_embeddings_test.py:317-333 in fp32"""
    state = [g * g for g in grads]
    new = [c - (g * np.float32(lr)) / (np.sqrt(s) + np.float32(eps)) for c, g, s in zip(cores, grads, state)]
    return new, state


def assert_adagrad_close(got_w, ref_w, ref_g, what="", lr=LR, eps=EPS, state0=None):
    """First Adagrad step w - lr*g/(sqrt(s0 + g^2)+eps): its derivative w.r.t. g is
    lr*(eps + s0-terms)/(|g|+eps)^2 <= lr/eps, so a gradient that is within the
    gradient tolerance (RTOL, ATOL_SCALE*max|g|) moves w by up to that factor.
    The gradient itself is checked at the tight tolerance by the dense-mode and
    optimizer-state tests."""
    got_w = np.asarray(got_w, dtype=np.float64)
    ref_w = np.asarray(ref_w, dtype=np.float64)
    g = np.abs(np.asarray(ref_g, dtype=np.float64))
    assert np.isfinite(got_w).all() and np.isfinite(ref_w).all(), f"{what}: non-finite values"
    dg = ATOL_SCALE * max(float(g.max()), 1e-30) + RTOL * g
    denom = (np.sqrt(g * g + (0.0 if state0 is None else np.asarray(state0, dtype=np.float64))) + eps)
    tol = RTOL * np.abs(ref_w) + 2e-7 * max(float(np.abs(ref_w).max()), 1e-30) + lr * dg * (eps + denom) / (denom * denom)
    err = np.abs(got_w - ref_w)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} out of tolerance; worst at {i}: got {got_w[i]!r} "
                             f"ref {ref_w[i]!r} tol {tol[i]:.3e} |g| {g[i]:.3e}")
