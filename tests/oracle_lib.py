"""ctypes binding of the CPU parity oracle (oracle/libttx_oracle.so).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/ttx_oracle.c.  Used by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_ROOT, "oracle", "libttx_oracle.so")

MAX_CORES = 4
OPTIM_SGD, OPTIM_ADAGRAD, OPTIM_DENSE = 0, 1, 2


class Geom(C.Structure):
    _fields_ = [
        ("T", C.c_int32),
        ("num_tables", C.c_int32),
        ("p", C.c_int32 * MAX_CORES),
        ("q", C.c_int32 * MAX_CORES),
        ("r", C.c_int32 * (MAX_CORES + 1)),
    ]


def make_geom(num_tables, p, q, ranks):
    """ranks: padded [1, r1, .., 1] or unpadded [r1, ..]."""
    T = len(p)
    ranks = list(ranks)
    if len(ranks) == T - 1:
        ranks = [1] + ranks + [1]
    assert len(ranks) == T + 1 and len(q) == T
    g = Geom()
    g.T = T
    g.num_tables = num_tables
    for t in range(T):
        g.p[t] = int(p[t])
        g.q[t] = int(q[t])
    for t in range(T + 1):
        g.r[t] = int(ranks[t])
    return g


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "libttx_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ttxo_hash64.restype = C.c_uint32
        _lib.ttxo_hash64.argtypes = [C.c_int64, C.c_int32]
        _lib.ttxo_hash64_raw.restype = C.c_uint32
        _lib.ttxo_hash64_raw.argtypes = [C.c_int64]
        _lib.ttxo_hash32.restype = C.c_uint32
        _lib.ttxo_hash32.argtypes = [C.c_int32, C.c_int32]
        _lib.ttxo_hashtbl_insert.restype = C.c_int32
        _lib.ttxo_hashtbl_insert.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        _lib.ttxo_hashtbl_find.restype = C.c_int32
        _lib.ttxo_hashtbl_find.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _ptrs(arrs):
    arr = (C.c_void_p * len(arrs))()
    for i, a in enumerate(arrs):
        arr[i] = a.ctypes.data
    return arr


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def hash64(key, size):
    return int(lib().ttxo_hash64(int(key), int(size)))


def hash64_raw(key):
    return int(lib().ttxo_hash64_raw(int(key)))


def hash32(key, size):
    return int(lib().ttxo_hash32(int(key), int(size)))


def hashtbl_insert(key, value, keys, values, max_probes=3):
    return int(lib().ttxo_hashtbl_insert(int(key), int(value), keys.size, max_probes, _p(keys), _p(values)))


def hashtbl_find(key, keys, max_probes=3):
    return int(lib().ttxo_hashtbl_find(int(key), keys.size, max_probes, _p(keys)))


def rowidx_from_offsets(offsets, num_tables):
    offsets = _i64(offsets)
    nb = offsets.size - 1
    B = nb // num_tables
    lengths = np.diff(offsets)
    bag = np.repeat(np.arange(nb, dtype=np.int64), lengths)
    return bag % B, bag // B


def tt_rows(geom, D, indices, tableidx, cores):
    indices = _i64(indices)
    cores = [_f32(c) for c in cores]
    rows = np.empty((indices.size, D), dtype=np.float32)
    tb = None if tableidx is None else _i64(tableidx)
    _check(lib().ttxo_tt_rows(C.byref(geom), C.c_int32(D), C.c_int64(indices.size), _p(indices), _p(tb), _ptrs(cores), _p(rows)), "tt_rows")
    return rows


def tt_forward(geom, B, D, indices, rowidx, tableidx, cores, nnz=None):
    indices, rowidx, tableidx = _i64(indices), _i64(rowidx), _i64(tableidx)
    nnz = indices.size if nnz is None else nnz
    cores = [_f32(c) for c in cores]
    out = np.empty((geom.num_tables, B, D), dtype=np.float32)
    _check(lib().ttxo_tt_forward(C.byref(geom), C.c_int32(B), C.c_int32(D), C.c_int64(nnz), _p(indices), _p(rowidx), _p(tableidx), _ptrs(cores), _p(out)), "tt_forward")
    return out


def tt_backward(geom, optim, B, D, lr, eps, indices, rowidx, tableidx, d_output, cores, state=None, nnz=None):
    """cores/state: lists of float32 arrays, updated IN PLACE for SGD/ADAGRAD.
    Returns the list of dense gradients for OPTIM_DENSE, else None."""
    indices, rowidx, tableidx = _i64(indices), _i64(rowidx), _i64(tableidx)
    nnz = indices.size if nnz is None else nnz
    d_output = _f32(d_output)
    for c in cores:
        assert c.dtype == np.float32 and c.flags["C_CONTIGUOUS"]
    d_cores = None
    dptr = None
    sptr = None
    if optim == OPTIM_DENSE:
        d_cores = [np.empty_like(c) for c in cores]
        dptr = _ptrs(d_cores)
    if optim == OPTIM_ADAGRAD:
        for s in state:
            assert s.dtype == np.float32 and s.flags["C_CONTIGUOUS"]
        sptr = _ptrs(state)
    _check(
        lib().ttxo_tt_backward(
            C.byref(geom), C.c_int32(optim), C.c_int32(B), C.c_int32(D), C.c_float(lr), C.c_float(eps), C.c_int64(nnz),
            _p(indices), _p(rowidx), _p(tableidx), _p(d_output), _ptrs(cores), sptr, dptr),
        "tt_backward",
    )
    return d_cores


def update_cache_state(indices, hashtbl, cache_freq):
    indices = _i64(indices)
    assert hashtbl.dtype == np.int64 and cache_freq.dtype == np.int64
    _check(lib().ttxo_update_cache_state(C.c_int64(indices.size), _p(indices), C.c_int64(hashtbl.size), _p(hashtbl), _p(cache_freq)), "update_cache_state")


def preprocess_indices(colidx, offsets, num_tables, warmup, hashtbl, cache_state):
    colidx, offsets = _i64(colidx), _i64(offsets)
    nnz = colidx.size
    rowidx = np.empty(nnz, dtype=np.int64)
    tableidx = np.empty(nnz, dtype=np.int64)
    pc = np.empty(nnz, dtype=np.int64)
    pr = np.empty(nnz, dtype=np.int64)
    pl = np.empty(nnz, dtype=np.int32)
    num_tt = C.c_int32(0)
    part = C.c_int32(0)
    _check(
        lib().ttxo_preprocess_indices(
            C.c_int64(nnz), _p(colidx), C.c_int64(offsets.size - 1), _p(offsets), C.c_int32(num_tables), C.c_int32(int(warmup)),
            C.c_int64(hashtbl.size), _p(hashtbl), _p(cache_state), _p(rowidx), _p(tableidx), _p(pc), _p(pr), _p(pl),
            C.byref(num_tt), C.byref(part)),
        "preprocess_indices",
    )
    if part.value:
        return pc, pr, tableidx, num_tt.value, pl
    return colidx, rowidx, tableidx, num_tt.value, None


def set_reference_exact(flag):
    """1: the oracle's cache_populate leaves the cache_state of evicted slots as the reference does (cu:1131-1133)"""
    lib().ttxo_set_reference_exact(C.c_int(int(flag)))


def cache_populate(geom, cores, hashtbl, cache_freq, cache_state, cache_weight):
    cores = [_f32(c) for c in cores]
    assert cache_weight.dtype == np.float32 and cache_state.dtype == np.int32
    _check(
        lib().ttxo_cache_populate(
            C.byref(geom), _ptrs(cores), C.c_int64(hashtbl.size), _p(hashtbl), _p(cache_freq), _p(cache_state),
            C.c_int64(cache_weight.shape[0]), C.c_int32(cache_weight.shape[1]), _p(cache_weight)),
        "cache_populate",
    )


def cache_forward(B, loc, rowidx, cache_weight, output):
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    rowidx = _i64(rowidx)
    _check(lib().ttxo_cache_forward(C.c_int32(B), C.c_int64(loc.size), _p(loc), _p(rowidx), C.c_int32(cache_weight.shape[1]), _p(cache_weight), _p(output)), "cache_forward")


def cache_backward_sgd(grad, loc, rowidx, lr, cache_weight):
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    rowidx, grad = _i64(rowidx), _f32(grad)
    _check(lib().ttxo_cache_backward_sgd(C.c_int64(loc.size), C.c_int32(cache_weight.shape[1]), _p(grad), _p(loc), _p(rowidx), C.c_float(lr), _p(cache_weight)), "cache_backward_sgd")


def cache_backward_dense(grad, loc, rowidx, cache_size, D):
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    rowidx, grad = _i64(rowidx), _f32(grad)
    out = np.empty((cache_size, D), dtype=np.float32)
    _check(lib().ttxo_cache_backward_dense(C.c_int64(loc.size), C.c_int32(D), _p(grad), _p(loc), _p(rowidx), C.c_int64(cache_size), _p(out)), "cache_backward_dense")
    return out


def cache_backward_rowwise_adagrad_approx(grad, loc, rowidx, lr, eps, state, cache_weight):
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    rowidx, grad = _i64(rowidx), _f32(grad)
    _check(
        lib().ttxo_cache_backward_rowwise_adagrad_approx(
            C.c_int64(loc.size), C.c_int32(cache_weight.shape[1]), _p(grad), _p(loc), _p(rowidx), C.c_float(lr), C.c_float(eps), _p(state), _p(cache_weight)),
        "cache_backward_rowwise_adagrad_approx",
    )


# ---- all-cores build (oracle/ttx_cpu_baseline.c): bench.py's cpu_baseline leg only -------------------------
_BASE_SO = os.path.join(_ROOT, "oracle", "libttx_cpu_baseline.so")
_blib = None


def baseline_lib():
    global _blib
    if _blib is None:
        if not os.path.exists(_BASE_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "libttx_cpu_baseline.so"])
        _blib = C.CDLL(_BASE_SO)
        _blib.ttxo_omp_threads.restype = C.c_int
    return _blib


def omp_threads():
    return int(baseline_lib().ttxo_omp_threads())


class OmpStep:
    """fwd + fused-optimizer bwd of one batch on all host cores (workspaces allocated once)"""

    def __init__(self, geom, B, D, nnz_max, cores):
        self.geom, self.B, self.D = geom, B, D
        self.threads = omp_threads()
        self.rows = np.empty((nnz_max, D), dtype=np.float32)
        self.grad = np.empty((self.threads, sum(int(c.size) for c in cores)), dtype=np.float32)
        self.out = np.empty((geom.num_tables, B, D), dtype=np.float32)

    def __call__(self, optim, lr, eps, indices, offsets, rowidx, tableidx, d_output, cores, state=None):
        indices, offsets, rowidx, tableidx = _i64(indices), _i64(offsets), _i64(rowidx), _i64(tableidx)
        d_output = _f32(d_output)
        _check(baseline_lib().ttxo_omp_step(
            C.byref(self.geom), C.c_int32(optim), C.c_int32(self.B), C.c_int32(self.D), C.c_float(lr), C.c_float(eps),
            C.c_int64(indices.size), _p(indices), _p(offsets), _p(rowidx), _p(tableidx), _p(d_output), _ptrs(cores),
            _ptrs(state) if state is not None else None, _p(self.out), _p(self.rows), _p(self.grad)), "omp_step")
        return self.out
