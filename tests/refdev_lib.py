"""ctypes binding of oracle/_ref/libcacheref.so: the REFERENCE's own hash-table templates and cache
kernels, compiled for gfx950 from the line ranges oracle/Makefile (target `refdev`) selects.

TEST INFRASTRUCTURE ONLY (GPU box).  All arguments are torch tensors on cuda:0; the product never imports
this module.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "oracle", "_ref", "libcacheref.so")
_lib = None


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
    return _lib


def _p(t):
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"reference kernel {what} failed: hipError {rc}")


def hash64(keys, H):
    out = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    _ok(lib().refdev_hash64(C.c_int(keys.numel()), _p(keys), C.c_int32(H), _p(out)), "hash64")
    return out.cpu().numpy().astype("uint32")


def insert_seq(keys, hashtbl, freq):
    """hashtbl_insert<int64,int64,true>(key, 1) key after key -> the return value of every insert"""
    ret = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    _ok(lib().refdev_insert_seq(C.c_int(keys.numel()), _p(keys), C.c_int32(hashtbl.numel()), _p(hashtbl), _p(freq), _p(ret)),
        "insert_seq")
    return ret


def find(keys, hashtbl):
    ret = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    _ok(lib().refdev_find(C.c_int(keys.numel()), _p(keys), C.c_int32(hashtbl.numel()), _p(hashtbl), _p(ret)), "find")
    return ret


def update_cache_state(colidx, hashtbl, freq, sequential=False):
    f = lib().refdev_update_cache_state_seq if sequential else lib().refdev_update_cache_state
    _ok(f(C.c_int(colidx.numel()), _p(colidx), C.c_int32(hashtbl.numel()), _p(hashtbl), _p(freq)), "update_cache_state")


def mark_popular(cache_size, sorted_keys, hashtbl, freq, cache_state):
    _ok(lib().refdev_mark_popular(C.c_int32(hashtbl.numel()), C.c_int32(cache_size), _p(sorted_keys), _p(hashtbl), _p(freq),
                                  _p(cache_state)), "mark_popular_colidx")


def compute_rowidx(offsets, num_tables, nnz):
    B = (offsets.numel() - 1) // num_tables
    rowidx = torch.empty(nnz, dtype=torch.int64, device=offsets.device)
    tableidx = torch.empty(nnz, dtype=torch.int64, device=offsets.device)
    _ok(lib().refdev_compute_rowidx(C.c_int32(B), C.c_int32(num_tables), _p(offsets), _p(rowidx), _p(tableidx)), "compute_rowidx")
    return rowidx, tableidx


def cache_lookup(colidx, hashtbl, cache_state):
    n = colidx.numel()
    is_tt = torch.zeros(n, dtype=torch.bool, device=colidx.device)
    loc = torch.full((n,), -1, dtype=torch.int32, device=colidx.device)  # (uninitialised for TT entries in the reference)
    _ok(lib().refdev_cache_lookup(C.c_int32(n), _p(colidx), C.c_int32(hashtbl.numel()), _p(hashtbl), _p(cache_state), _p(is_tt),
                                  _p(loc)), "cache_lookup")
    return is_tt, loc


def cache_forward(rowidx, loc, cache_weight, output):
    _ok(lib().refdev_cache_forward(C.c_int32(loc.numel()), C.c_int32(cache_weight.size(1)), _p(rowidx), _p(loc), _p(cache_weight),
                                   _p(output)), "cache_forward")


def cache_backward_sgd(grad, loc, rowidx, lr, cache_weight):
    _ok(lib().refdev_cache_backward_sgd(C.c_int32(loc.numel()), C.c_int32(cache_weight.size(1)), _p(grad), _p(loc), _p(rowidx),
                                        C.c_float(lr), _p(cache_weight)), "cache_backward_sgd")


def cache_backward_dense(grad, loc, rowidx, cache_size):
    D = grad.size(-1)
    out = torch.zeros(cache_size, D, dtype=torch.float32, device=grad.device)
    _ok(lib().refdev_cache_backward_dense(C.c_int32(loc.numel()), C.c_int32(D), _p(grad), _p(loc), _p(rowidx), _p(out)),
        "cache_backward_dense")
    return out
