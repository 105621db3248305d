"""Rows a6 / a13 pinned to a LIBRARY's statement of the CUB contract, not only to a paragraph restating it: the
product's own stable partition (csrc/ttx_cache.hip rowidx_update / partition_scatter; reference: four
cub::DevicePartition::Flagged calls, tt_embeddings_cuda.cu:1437-1478) and its 64-bit stable descending radix sort
(radix_*; reference: cub::DeviceRadixSort::SortPairsDescending, cu:1280-1308) against hipCUB's implementation of those
very calls (oracle/primref.hip -> oracle/libprimref.so, test infrastructure).  Bit-exact, 1 .. 2^20 items, all selected /
none selected / frequency ties."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libprimref.so")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def prim():
    assert os.path.exists(_SO), "oracle/libprimref.so not built: run __graft_entry__.build()"
    return C.CDLL(_SO)


def hipcub_partition(x, flags):
    n = x.numel()
    out = torch.empty_like(x)
    nsel = torch.zeros(1, dtype=torch.int32, device=DEV)
    f = prim().primref_partition_flagged_i64 if x.dtype == torch.int64 else prim().primref_partition_flagged_i32
    rc = f(C.c_void_p(x.data_ptr()), C.c_void_p(flags.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(nsel.data_ptr()), C.c_int(n))
    assert rc == 0, f"hipcub::DevicePartition::Flagged failed: {rc}"
    return out, int(nsel.item())


@pytest.mark.parametrize("nnz,pattern", [(1, "mixed"), (63, "mixed"), (64, "all"), (257, "none"), (1000, "mixed"), (4096, "all"), (20000, "none"),
                                         (70001, "mixed"), (1 << 20, "mixed"), (1 << 20, "sparse")])
def test_partition_equals_hipcub_flagged(nnz, pattern):
    """preprocess_indices_sync(warmup=False): TT entries first in index order, cached entries behind them REVERSED --
    three arrays, as the reference partitions colidx, rowidx and cache_locations with the same flags"""
    import tt_embeddings as E

    rs = np.random.RandomState(nnz % 9973 + len(pattern))
    H, E_ = 1 << 16, 50_000
    keys, freq = np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)
    O.update_cache_state(rs.randint(0, E_, size=30000).astype(np.int64), keys, freq)
    present = keys != -1
    frac = {"mixed": 0.5, "all": 0.0, "none": 1.0, "sparse": 0.02}[pattern]  # share of the table's keys that are cached
    state = np.where(present & (rs.rand(H) < frac), rs.randint(0, 4000, size=H), -1).astype(np.int32)
    if pattern == "none":  # every index of the batch is cached: draw the batch from the cached keys
        idx = rs.choice(keys[present], size=nnz).astype(np.int64)
    elif pattern == "all":
        idx = rs.randint(E_, 2 * E_, size=nnz).astype(np.int64)  # keys the table has never seen
    else:
        idx = rs.randint(0, E_, size=nnz).astype(np.int64)
    B = max(1, nnz // 7)
    lens = rs.multinomial(nnz, np.ones(B) / B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ti, to = t(idx), t(off)
    pc, pr, _, ntt, ploc = E.preprocess_indices_sync(ti, to, 1, False, t(keys), t(state))
    # the flags and the unpartitioned arrays, from the oracle's restatement of cache_lookup_kernel (bit-exact vs the
    # reference's kernel: tests/test_refdev_gpu.py)
    rowidx, _ = O.rowidx_from_offsets(off, 1)
    if nnz <= 70001:
        loc = np.array([state[s] if s >= 0 else -1 for s in (O.hashtbl_find(int(k), keys) for k in idx)], dtype=np.int32)
    else:  # (2^20 items: the reference's own cache_lookup_kernel, compiled for gfx950 -- oracle/_ref/libcacheref.so)
        import refdev_lib as R

        if not R.available():
            pytest.skip("oracle/_ref/libcacheref.so not built")
        loc = R.cache_lookup(ti, t(keys), t(state))[1].cpu().numpy().astype(np.int32)
    flags = t(loc == -1)
    ec, n1 = hipcub_partition(ti, flags)
    er, n2 = hipcub_partition(t(rowidx), flags)
    el, n3 = hipcub_partition(t(loc), flags)
    assert n1 == n2 == n3 == ntt == int((loc == -1).sum())
    if pattern == "all":
        assert ntt == nnz
    if pattern == "none":
        assert ntt == 0
    assert torch.equal(pc, ec), "partitioned colidx"
    assert torch.equal(pr, er), "partitioned rowidx"
    if ntt < nnz:
        assert torch.equal(ploc[ntt:], el[ntt:]), "partitioned cache locations (the rejected, reversed)"


@pytest.mark.parametrize("n,kind", [(1, "ties"), (2, "distinct"), (255, "ties"), (256, "zipf"), (4097, "zipf"), (1 << 16, "ties"),
                                    (1 << 20, "zipf"), (1 << 20, "wide"), (300000, "allzero")])
def test_sort_equals_hipcub_sort_pairs_descending(n, kind):
    """cache_populate's sort of (cache_freq, hashtbl): descending and STABLE (equal frequencies keep ascending slot order)"""
    import tt_embeddings as E

    rs = np.random.RandomState(n % 7919)
    if kind == "ties":
        k = rs.randint(0, 4, size=n)
    elif kind == "distinct":
        k = rs.permutation(n) + 1
    elif kind == "zipf":
        k = np.minimum(rs.zipf(1.3, size=n), 1 << 40)
    elif kind == "wide":
        k = rs.randint(0, 1 << 62, size=n)
    else:
        k = np.zeros(n)
    k = k.astype(np.int64)
    v = rs.randint(-1, 1 << 40, size=n).astype(np.int64)
    tk, tv = t(k), t(v)
    gk, gv = E.debug_sort_pairs_desc(tk, tv)
    ek, ev = torch.empty_like(tk), torch.empty_like(tv)
    rc = prim().primref_sort_pairs_desc_i64(C.c_void_p(tk.data_ptr()), C.c_void_p(ek.data_ptr()), C.c_void_p(tv.data_ptr()),
                                            C.c_void_p(ev.data_ptr()), C.c_int(n))
    assert rc == 0, f"hipcub::DeviceRadixSort::SortPairsDescending failed: {rc}"
    assert torch.equal(gk, ek), "sorted keys"
    assert torch.equal(gv, ev), "sorted values (stability)"
    order = np.argsort(-k, kind="stable")
    assert np.array_equal(gv.cpu().numpy(), v[order]), "the documented contract: stable, descending"
