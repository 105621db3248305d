"""`tt_embeddings` -- drop-in for the reference's native extension module.

The reference builds a pybind11/CUDA extension called `tt_embeddings` exporting
eleven functions (reference tt_embeddings.cpp:131-161).  This module exports the
same eleven names with the same positional arguments, tensor conventions and
error behaviour (RuntimeError where the reference's TORCH_CHECK fires), and
forwards every call to the hand-written HIP library `libttx.so` (gfx950) through
its C ABI (include/ttx.h) with ctypes.  PyTorch is used only for device memory
and the current HIP stream.

There is NO CPU or PyTorch fallback: if `libttx.so` is missing, or a tensor is
not on a GPU, the call raises.

Extras beyond the reference's eleven names (used by tt_embeddings_ops.py and
bench.py): `make_plan` (share the lookup plan between forward and backward; `dedup=True`: duplicate lookups of
the batch share one contraction),
`profile_*` (live kernel timings), `lib()` (the loaded ctypes library).
"""
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("TTX_LIB") or os.path.join(_HERE, "libttx.so")  # TTX_LIB: try another build of the same library

MAX_CORES = 4
OPTIM_SGD, OPTIM_ADAGRAD, OPTIM_DENSE = 0, 1, 2
PROF_FWD, PROF_BWD, PROF_APPLY, PROF_PLAN, PROF_POOL, PROF_CACHE_FWD = range(6)
FUSED_CACHE_UPDATE = True  # preprocess_indices_sync(..., update_cache_freq=) exists


class _Geom(C.Structure):
    _fields_ = [
        ("T", C.c_int32),
        ("num_tables", C.c_int32),
        ("p", C.c_int32 * MAX_CORES),
        ("q", C.c_int32 * MAX_CORES),
        ("r", C.c_int32 * (MAX_CORES + 1)),
        ("p_tables", C.POINTER(C.c_int32)),  # NULL, or [num_tables][T]: tables of different row factors
    ]


_lib = None        # the library calls go to: the product build, or the test build while a knob is away from its default
_product = None    # libttx.so
_hooks = None      # libttx_hooks.so (same sources, -DTTX_TEST_HOOKS): loaded on the first use of a test / ablation knob
_SO_HOOKS = os.environ.get("TTX_LIB_HOOKS") or os.path.join(_HERE, "libttx_hooks.so")


def lib():
    """libttx.so, loaded once.  Raises RuntimeError if it has not been built.  (While a test / ablation knob is away from
    its default -- debug_skip(), set_chunk(), debug_lds_budget(), ... below -- calls go to libttx_hooks.so instead: the
    product library has no knobs.)"""
    global _lib, _product
    if _lib is not None:
        return _lib
    _product = _load(_SO)
    _lib = _product
    if os.environ.get("TTX_LDS_BUDGET"):  # experiments: LDS budget of the generic kernels' tile search (bytes)
        debug_lds_budget(int(os.environ["TTX_LDS_BUDGET"]))
    if os.environ.get("TTX_BWD32"):  # experiment (A/B scripts): bwd32_kernel with this many lookups per chunk (1 = 128)
        debug_bwd32(int(os.environ.get("TTX_BWD32_MC") or 128) if os.environ["TTX_BWD32"] == "1" else int(os.environ["TTX_BWD32"]))
    if os.environ.get("TTX_DEBUG_SKIP"):  # ablation runs (scripts/upper_bounds.py): results INVALID while set
        debug_skip(int(os.environ["TTX_DEBUG_SKIP"]))
    return _lib


def hooks_lib():
    """libttx_hooks.so (the test build: include/ttx_test_hooks.h), loaded once; calls are routed to it from now on and back
    to the product library as soon as every knob is at its default again (`_knobs_changed`)."""
    global _lib, _hooks
    lib()
    if _hooks is None:
        _hooks = _load(_SO_HOOKS)
        if not _hooks.ttx_has_test_hooks():
            raise RuntimeError(f"tt_embeddings: {_SO_HOOKS} was built without -DTTX_TEST_HOOKS")
        i32 = C.c_int32
        _hooks.ttx_set_chunk.argtypes = [i32]
        _hooks.ttx_debug_lds_budget.argtypes = [i32]
        _hooks.ttx_debug_skip.argtypes = [i32]
        _hooks.ttx_debug_cache_fwd.argtypes = [i32]
        _hooks.ttx_debug_stamps.argtypes = [C.c_void_p]
        _hooks.ttx_debug_bwd32.argtypes = [i32]
    _lib = _hooks
    return _hooks


def _knobs_changed() -> None:
    global _lib
    if _hooks is not None and _lib is _hooks and not _hooks.ttx_debug_state():
        _lib = _product


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"tt_embeddings: the HIP library {path} is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU / PyTorch fallback."
        )
    L = C.CDLL(path)
    L.ttx_last_error.restype = C.c_char_p
    for name in (
        "ttx_plan_bytes",
        "ttx_tt_forward_workspace_bytes",
        "ttx_tt_backward_workspace_bytes",
        "ttx_preprocess_workspace_bytes",
        "ttx_cache_populate_workspace_bytes",
    ):
        getattr(L, name).restype = C.c_size_t
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    G = C.POINTER(_Geom)
    L.ttx_plan_bytes.argtypes = [G, i64]
    L.ttx_plan_build.argtypes = [G, i64, vp, vp, vp, vp, sz, vp]
    L.ttx_tt_forward_workspace_bytes.argtypes = [G, i32, i32, i64]
    L.ttx_tt_forward.argtypes = [G, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_tt_forward_o.argtypes = [G, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_tt_forward_arrive_ints.argtypes = [G, i64]
    L.ttx_tt_forward_arrive_ints.restype = i64
    L.ttx_tt_rows.argtypes = [G, i32, i64, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_tt_backward_workspace_bytes.argtypes = [G, i32, i32, i64]
    L.ttx_tt_backward.argtypes = [G, i32, i32, i32, f32, f32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_split0_expand.argtypes = [i64, i64, i32, i64, vp, vp, vp, vp, vp]
    L.ttx_tt_backward_w.argtypes = [G, i32, i32, i32, f32, f32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_update_cache_state.argtypes = [i64, vp, i64, vp, vp, vp]
    L.ttx_preprocess_workspace_bytes.argtypes = [i64]
    L.ttx_preprocess_indices_sync.argtypes = [i64, vp, i64, vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp,
                                              C.POINTER(i32), C.POINTER(i32), vp, sz, vp]
    L.ttx_preprocess_indices_sync_fused.argtypes = [i64, vp, i64, vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp,
                                                    C.POINTER(i32), C.POINTER(i32), vp, vp, vp, sz, vp]
    L.ttx_lookup_prologue.argtypes = [C.POINTER(_Geom), i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_cache_populate_workspace_bytes.argtypes = [G, i64, i64, i32]
    L.ttx_cache_populate.argtypes = [G, vp, i64, vp, vp, vp, i64, i32, vp, vp, sz, vp]
    L.ttx_cache_forward.argtypes = [i32, i64, vp, vp, i32, vp, vp, vp]
    L.ttx_cache_backward_sgd.argtypes = [i64, i32, vp, vp, vp, f32, vp, vp]
    L.ttx_cache_backward_dense.argtypes = [i64, i32, vp, vp, vp, i64, vp, vp]
    L.ttx_cache_backward_rowwise_adagrad_approx.argtypes = [i64, i32, vp, vp, vp, f32, f32, vp, vp, vp]
    L.ttx_profile_enable.argtypes = [C.c_int]
    L.ttx_profile_read.argtypes = [C.c_int, C.POINTER(i64), C.POINTER(C.c_double)]
    L.ttx_debug_state.argtypes = []
    L.ttx_debug_state.restype = C.c_int
    L.ttx_has_test_hooks.argtypes = []
    L.ttx_debug_tiles.argtypes = [G, C.POINTER(i32)]
    L.ttx_cache_populate_f.argtypes = [G, vp, i64, vp, vp, vp, i64, i32, vp, i32, vp, sz, vp]
    L.ttx_cache_backward_sorted_workspace_bytes.restype = C.c_size_t
    L.ttx_cache_backward_sorted_workspace_bytes.argtypes = [i64, i64, i32]
    L.ttx_cache_backward_sorted.argtypes = [i32, i64, vp, i64, i32, vp, vp, vp, f32, f32, i64, vp, vp, vp, sz, vp]
    for name in ("ttx_dedup_bytes", "ttx_tt_forward_dd_workspace_bytes", "ttx_tt_backward_dd_workspace_bytes"):
        getattr(L, name).restype = C.c_size_t
    L.ttx_dedup_bytes.argtypes = [G, i64]
    L.ttx_dedup_build.argtypes = [G, i64, vp, vp, vp, sz, vp, sz, vp]
    L.ttx_tt_forward_dd_workspace_bytes.argtypes = [G, i32, i64]
    L.ttx_tt_forward_dd.argtypes = [G, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.ttx_tt_backward_dd_workspace_bytes.argtypes = [G, i32, i64]
    L.ttx_tt_backward_dd.argtypes = [G, i32, i32, i32, f32, f32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("tt_embeddings (libttx): " + lib().ttx_last_error().decode())


_geom_cache = {}


def _geom(num_tables: int, p: Sequence, q: Sequence[int], ranks: Sequence[int]) -> _Geom:
    """`p` = tt_p_shapes, or (beyond the reference: include/ttx.h ttx_geom::p_tables) one list per table for
    tables of different row factors -- the cores are then [1, sum of the tables' p_t, slice]."""
    if len(p) > 0 and isinstance(p[0], (list, tuple)):
        return _geom_mixed(p, q, ranks)
    key = (int(num_tables), tuple(int(x) for x in p), tuple(int(x) for x in q), tuple(int(x) for x in ranks))
    g = _geom_cache.get(key)
    if g is None:
        T = len(key[1])
        if not (2 <= T <= MAX_CORES) or len(key[2]) != T or len(key[3]) != T + 1:
            raise RuntimeError(
                f"tt_embeddings: need 2..4 cores with len(q) == len(p) and len(ranks) == len(p)+1, got "
                f"p={list(key[1])} q={list(key[2])} ranks={list(key[3])}")
        g = _Geom()
        g.T, g.num_tables = T, key[0]
        for t in range(T):
            g.p[t], g.q[t] = key[1][t], key[2][t]
        for t in range(T + 1):
            g.r[t] = key[3][t]
        _geom_cache[key] = g
    return g


def _geom_mixed(p, q, ranks) -> _Geom:
    key = ("mixed", tuple(tuple(int(x) for x in row) for row in p), tuple(int(x) for x in q), tuple(int(x) for x in ranks))
    g = _geom_cache.get(key)
    if g is None:
        T = len(key[2])
        if not (2 <= T <= MAX_CORES) or len(key[3]) != T + 1 or any(len(row) != T for row in key[1]):
            raise RuntimeError(f"tt_embeddings: per-table tt_p_shapes need {T} factors each and len(ranks) == {T + 1}")
        g = _Geom()
        g.T, g.num_tables = T, len(key[1])
        flat = [v for row in key[1] for v in row]
        g._ptab = (C.c_int32 * len(flat))(*flat)  # kept alive by the cached struct
        g.p_tables = C.cast(g._ptab, C.POINTER(C.c_int32))
        g._psum = [sum(row[t] for row in key[1]) for t in range(T)]
        for t in range(T):
            g.p[t], g.q[t] = max(row[t] for row in key[1]), key[2][t]
        for t in range(T + 1):
            g.r[t] = key[3][t]
        _geom_cache[key] = g
    return g


def _dev(t: torch.Tensor) -> torch.device:
    if not t.is_cuda:
        raise RuntimeError("tt_embeddings: tensors must live on a GPU (no CPU path in this build)")
    return t.device


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev: torch.device) -> int:
    """the current HIP stream of `dev` as an integer handle (the raw getter skips building a
    torch.cuda.Stream object: ~8 us per call on the lookup path)"""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _guard:
    """OptionalCUDAGuard of the reference entry points (e.g. cu:436-437)."""

    __slots__ = ("dev", "prev")

    def __init__(self, dev):
        self.dev = dev.index if dev.index is not None else torch.cuda.current_device()
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev:
            self.prev = cur
            torch.cuda.set_device(self.dev)

    def __exit__(self, *a):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


_ws_cache = {}


def _workspace(dev: torch.device, stream: int, nbytes: int) -> torch.Tensor:
    """Grow-only scratch per (device, stream): stream order makes reuse safe."""
    key = (dev.index, stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws


_arrive_cache = {}


def _arrive_zeros(dev: torch.device, stream: int, n: int) -> torch.Tensor:
    """the arrival counters of fused pooling: zero before and after every call, so one array per (device, stream)"""
    key = (dev.index, stream)
    t = _arrive_cache.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1 << 16), dtype=torch.int32, device=dev)
        _arrive_cache[key] = t
    return t


def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _i64(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.int64:
        raise RuntimeError(f"tt_embeddings: {name} must be int64, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"tt_embeddings: {name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _cores(tt_cores: Sequence[torch.Tensor], g: _Geom, name="tt_cores"):
    if len(tt_cores) != g.T:
        raise RuntimeError(f"tt_embeddings: expected {g.T} {name}, got {len(tt_cores)}")
    out = []
    for t, c in enumerate(tt_cores):
        c = c.detach() if c.requires_grad else c
        want = (g.num_tables, g.p[t], g.r[t] * g.q[t] * g.r[t + 1])
        if bool(g.p_tables):  # tables of different row factors: one array of all their slices
            want = (1, g._psum[t], want[2])
        if tuple(c.shape) != want or c.dtype != torch.float32 or not c.is_contiguous() or not c.is_cuda:
            raise RuntimeError(f"tt_embeddings: {name}[{t}] must be a contiguous float32 GPU tensor of shape {want}, "
                               f"got {tuple(c.shape)} {c.dtype} on {c.device}")
        out.append(c)
    return out


class Plan:
    """Device-resident lookup plan shared by forward and backward of one batch."""

    __slots__ = ("buf", "nnz", "key")

    def __init__(self, buf, nnz, key):
        self.buf, self.nnz, self.key = buf, nnz, key


class DedupPlan(Plan):
    """A plan of the batch's DISTINCT (table, index) pairs plus the map of the lookups onto them (include/ttx.h,
    "duplicate lookups"): tt_forward / the backward entry points given such a plan contract every distinct pair
    once.  Built by make_plan(..., dedup=True) when the batch qualifies."""

    __slots__ = ("dd",)

    def __init__(self, buf, dd, nnz, key):
        super().__init__(buf, nnz, key)
        self.dd = dd


def make_plan(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks, nnz, indices, tableidx, rowidx=None,
              dedup: bool = False) -> Optional[Plan]:
    if nnz == 0:
        return None
    g = _geom(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks)
    dev = _dev(indices)
    indices, tableidx = _i64(indices, "indices"), _i64(tableidx, "tableidx")
    L = lib()
    if dedup:
        db = L.ttx_dedup_bytes(C.byref(g), nnz)
        if db:  # (0: more than 16384 lookups or a key space beyond 2^32 -- the plain plan gives the same results)
            nb = L.ttx_plan_bytes(C.byref(g), nnz)
            buf = torch.empty(nb, dtype=torch.uint8, device=dev)
            dd = torch.empty(db, dtype=torch.uint8, device=dev)
            with _guard(dev):
                _check(L.ttx_dedup_build(C.byref(g), nnz, indices.data_ptr(), tableidx.data_ptr(), dd.data_ptr(), db,
                                         buf.data_ptr(), nb, _stream(dev)))
            return DedupPlan(buf, dd, nnz, (num_tables, tuple(tt_p_shapes), tuple(tt_q_shapes), tuple(tt_ranks)))
    nb = L.ttx_plan_bytes(C.byref(g), nnz)
    buf = torch.empty(nb, dtype=torch.uint8, device=dev)
    with _guard(dev):
        _check(L.ttx_plan_build(C.byref(g), nnz, indices.data_ptr(), tableidx.data_ptr(),
                                None if rowidx is None else _i64(rowidx, "rowidx").data_ptr(), buf.data_ptr(), nb, _stream(dev)))
    return Plan(buf, nnz, (num_tables, tuple(tt_p_shapes), tuple(tt_q_shapes), tuple(tt_ranks)))


def split0_expand(indices: torch.Tensor, offsets: torch.Tensor, k: int, p_rest: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Not in the reference: the part lookups of a table whose first factor is split k ways (include/ttx.h, "core-0 row
    split").  offsets: nb + 1 entries.  -> (indices [k nnz], offsets [k nb + 1])."""
    dev = _dev(indices)
    indices, offsets = _i64(indices, "indices"), _i64(offsets, "offsets")
    nnz, nb = indices.numel(), offsets.numel() - 1
    out_i = torch.empty(k * nnz, dtype=torch.int64, device=dev)
    out_o = torch.empty(k * nb + 1, dtype=torch.int64, device=dev)
    with _guard(dev):
        _check(lib().ttx_split0_expand(nnz, nb, k, p_rest, indices.data_ptr(), offsets.data_ptr(), out_i.data_ptr(),
                                       out_o.data_ptr(), _stream(dev)))
    return out_i, out_o


def lookup_prologue(colidx: torch.Tensor, offsets: torch.Tensor, num_tables: int, tt_p_shapes, tt_q_shapes, tt_ranks,
                    hashtbl: Optional[torch.Tensor] = None, cache_freq: Optional[torch.Tensor] = None
                    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[Plan]]:
    """Not in the reference: update_cache_state (when hashtbl / cache_freq are given) +
    preprocess_indices_sync(warmup=True) + make_plan of one batch in ONE native call (one launch
    when the batch qualifies, include/ttx.h).  -> (rowidx, tableidx, plan)."""
    dev = _dev(colidx)
    colidx, offsets = _i64(colidx, "colidx"), _i64(offsets, "offsets")
    nnz = colidx.numel()
    rowidx = torch.empty_like(colidx)
    tableidx = torch.empty_like(colidx)
    if nnz == 0:
        return rowidx, tableidx, None
    g = _geom(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks)
    upd = hashtbl is not None and cache_freq is not None and hashtbl.numel() > 0
    if upd and hashtbl.numel() != cache_freq.numel():
        raise RuntimeError("tt_embeddings: hashtbl must match cache_freq")
    L = lib()
    nb = L.ttx_plan_bytes(C.byref(g), nnz)
    buf = torch.empty(nb, dtype=torch.uint8, device=dev)
    with _guard(dev):
        _check(L.ttx_lookup_prologue(C.byref(g), nnz, colidx.data_ptr(), offsets.numel() - 1, offsets.data_ptr(),
                                     hashtbl.numel() if upd else 0, hashtbl.data_ptr() if upd else None,
                                     cache_freq.data_ptr() if upd else None, rowidx.data_ptr(), tableidx.data_ptr(),
                                     buf.data_ptr(), nb, _stream(dev)))
    return rowidx, tableidx, Plan(buf, nnz, (num_tables, tuple(tt_p_shapes), tuple(tt_q_shapes), tuple(tt_ranks)))


def _plan_ptr(plan: Optional[Plan], nnz: int):
    if plan is None:
        return None
    if plan.nnz != nnz:
        raise RuntimeError("tt_embeddings: plan was built for a different nnz")
    return plan.buf.data_ptr()


# --------------------------------------------------------------------------- #
# the eleven reference entry points (tt_embeddings.cpp:131-161)
# --------------------------------------------------------------------------- #

def tt_forward(batch_count: int, num_tables: int, B: int, D: int, tt_p_shapes: List[int], tt_q_shapes: List[int],
               tt_ranks: List[int], L: torch.Tensor, nnz: int, indices: torch.Tensor, rowidx: torch.Tensor,
               tableidx: torch.Tensor, tt_cores: List[torch.Tensor], plan: Optional[Plan] = None,
               offsets: Optional[torch.Tensor] = None, per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tt_embeddings.cpp:13-26.  `batch_count` (the reference's GEMM chunk size)
    is accepted and ignored: the HIP path has no chunk loop.  `L` is validated
    for length only; strides are derived from tt_p_shapes."""
    g = _geom(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks)
    dev = _dev(tt_cores[0])
    cores = _cores(tt_cores, g)
    out = torch.empty((num_tables, B, D), dtype=torch.float32, device=dev)
    if nnz > 0 and batch_count <= 0:
        raise RuntimeError("tt_embeddings: batch_count must be > 0")  # cu:987
    if L.numel() != g.T:
        raise RuntimeError("tt_embeddings: L must have one stride per core")
    indices, rowidx, tableidx = _i64(indices, "indices"), _i64(rowidx, "rowidx"), _i64(tableidx, "tableidx")
    if nnz > indices.numel() or nnz > rowidx.numel() or nnz > tableidx.numel():
        raise RuntimeError("tt_embeddings: nnz exceeds the index tensors")
    lb = lib()
    st = _stream(dev)
    if isinstance(plan, DedupPlan) and nnz > 0:
        nb = lb.ttx_tt_forward_dd_workspace_bytes(C.byref(g), D, nnz)
        ws = _workspace(dev, st, nb)
        with _guard(dev):
            psw = None if per_sample_weights is None else _f32(per_sample_weights, "per_sample_weights")
            _check(lb.ttx_tt_forward_dd(C.byref(g), B, D, nnz, rowidx.data_ptr(), tableidx.data_ptr(),
                                        None if psw is None else psw.data_ptr(), plan.dd.data_ptr(),
                                        _plan_ptr(plan, nnz), _ptr_array(cores), out.data_ptr(), ws.data_ptr(), ws.numel(), st))
        return out
    nb = lb.ttx_tt_forward_workspace_bytes(C.byref(g), B, D, nnz)
    ws = _workspace(dev, st, nb)
    if offsets is not None or per_sample_weights is not None:
        # beyond the reference's signature: the bags' offsets (the ones rowidx / tableidx were derived from) let the
        # contraction kernel pool the bags itself (include/ttx.h ttx_tt_forward_o) -- same output bit for bit
        na = lb.ttx_tt_forward_arrive_ints(C.byref(g), nnz) if (offsets is not None and nnz > 0) else 0
        arrive = _arrive_zeros(dev, st, na) if na > 0 else None
        psw = None if per_sample_weights is None else _f32(per_sample_weights, "per_sample_weights")
        with _guard(dev):
            _check(lb.ttx_tt_forward_o(C.byref(g), B, D, nnz, indices.data_ptr(), rowidx.data_ptr(), tableidx.data_ptr(),
                                       None if psw is None else psw.data_ptr(), _ptr_array(cores), out.data_ptr(), None,
                                       _i64(offsets, "offsets").data_ptr() if na > 0 else None,
                                       arrive.data_ptr() if na > 0 else None, _plan_ptr(plan, nnz), ws.data_ptr(), ws.numel(), st))
        return out
    with _guard(dev):
        _check(lb.ttx_tt_forward(C.byref(g), B, D, nnz, indices.data_ptr(), rowidx.data_ptr(), tableidx.data_ptr(),
                                 _ptr_array(cores), out.data_ptr(), _plan_ptr(plan, nnz), ws.data_ptr(), ws.numel(), st))
    return out


def _backward(optim, D, lr, eps, p, q, ranks, nnz, indices, rowidx, tableidx, d_output, tt_cores, state, plan,
              per_sample_weights=None):
    g = _geom(tt_cores[0].size(0), p, q, ranks)
    num_tables = g.num_tables  # (tables of different row factors: len(p), the cores are [1, sum p, slice])
    dev = _dev(d_output)
    cores = _cores(tt_cores, g)
    d_output = _f32(d_output, "d_output")
    if d_output.dim() != 3 or d_output.size(0) != num_tables or d_output.size(2) != D:
        raise RuntimeError(f"tt_embeddings: d_output must be [num_tables, B, D], got {tuple(d_output.shape)}")
    B = d_output.size(1)
    indices, rowidx, tableidx = _i64(indices, "indices"), _i64(rowidx, "rowidx"), _i64(tableidx, "tableidx")
    grads = None
    gptr = sptr = None
    if optim == OPTIM_DENSE:
        grads = [torch.empty_like(c) for c in cores]
        gptr = _ptr_array(grads)
    if optim == OPTIM_ADAGRAD:
        sptr = _ptr_array(_cores(state, g, "optimizer_state"))
    lb = lib()
    st = _stream(dev)
    # (beyond the reference's signature: nn.EmbeddingBag's per_sample_weights scale each lookup's share of its bag gradient)
    psw = None if per_sample_weights is None else _f32(per_sample_weights, "per_sample_weights")
    pswp = None if psw is None else psw.data_ptr()
    if isinstance(plan, DedupPlan) and nnz > 0:
        nb = lb.ttx_tt_backward_dd_workspace_bytes(C.byref(g), D, nnz)
        ws = _workspace(dev, st, nb)
        with _guard(dev):
            _check(lb.ttx_tt_backward_dd(C.byref(g), optim, B, D, lr, eps, nnz, rowidx.data_ptr(), tableidx.data_ptr(), pswp,
                                         d_output.data_ptr(), plan.dd.data_ptr(), _plan_ptr(plan, nnz), _ptr_array(cores), sptr,
                                         gptr, ws.data_ptr(), ws.numel(), st))
        return grads
    nb = lb.ttx_tt_backward_workspace_bytes(C.byref(g), B, D, nnz)
    ws = _workspace(dev, st, nb)
    with _guard(dev):
        if psw is not None:
            _check(lb.ttx_tt_backward_w(C.byref(g), optim, B, D, lr, eps, nnz, indices.data_ptr(), rowidx.data_ptr(),
                                        tableidx.data_ptr(), pswp, d_output.data_ptr(), _ptr_array(cores), sptr, gptr,
                                        _plan_ptr(plan, nnz), ws.data_ptr(), ws.numel(), st))
        else:
            _check(lb.ttx_tt_backward(C.byref(g), optim, B, D, lr, eps, nnz, indices.data_ptr(), rowidx.data_ptr(),
                                      tableidx.data_ptr(), d_output.data_ptr(), _ptr_array(cores), sptr, gptr,
                                      _plan_ptr(plan, nnz), ws.data_ptr(), ws.numel(), st))
    return grads


def tt_dense_backward(batch_count, D, tt_p_shapes, tt_q_shapes, tt_ranks, L, nnz, indices, rowidx, tableidx, d_output,
                      tt_cores, plan: Optional[Plan] = None, per_sample_weights: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """tt_embeddings.cpp:28-40: returns the list of dense core gradients."""
    return _backward(OPTIM_DENSE, D, 0.0, 0.0, tt_p_shapes, tt_q_shapes, tt_ranks, nnz, indices, rowidx, tableidx,
                     d_output, tt_cores, None, plan, per_sample_weights)


def tt_sgd_backward(batch_count, D, learning_rate, tt_p_shapes, tt_q_shapes, tt_ranks, L, nnz, indices, rowidx,
                    tableidx, d_output, tt_cores, plan: Optional[Plan] = None,
                    per_sample_weights: Optional[torch.Tensor] = None) -> None:
    """tt_embeddings.cpp:42-55: fused gradient + SGD, cores updated in place."""
    _backward(OPTIM_SGD, D, learning_rate, 0.0, tt_p_shapes, tt_q_shapes, tt_ranks, nnz, indices, rowidx, tableidx,
              d_output, tt_cores, None, plan, per_sample_weights)


def tt_adagrad_backward(batch_count, D, learning_rate, eps, tt_p_shapes, tt_q_shapes, tt_ranks, L, nnz, indices,
                        rowidx, tableidx, d_output, optimizer_state, tt_cores, plan: Optional[Plan] = None,
                        per_sample_weights: Optional[torch.Tensor] = None) -> None:
    """tt_embeddings.cpp:57-72: fused gradient + Adagrad, cores and state in place."""
    _backward(OPTIM_ADAGRAD, D, learning_rate, eps, tt_p_shapes, tt_q_shapes, tt_ranks, nnz, indices, rowidx,
              tableidx, d_output, tt_cores, list(optimizer_state), plan, per_sample_weights)


def update_cache_state(indices: torch.Tensor, hashtbl: torch.Tensor, cache_freq: torch.Tensor) -> None:
    """tt_embeddings.cpp:74."""
    nnz = indices.numel()
    if nnz == 0:
        return
    dev = _dev(indices)
    if hashtbl.numel() <= 0 or hashtbl.numel() != cache_freq.numel():  # cu:1099-1100
        raise RuntimeError("tt_embeddings: hashtbl must be non-empty and match cache_freq")
    indices = _i64(indices, "indices")
    _i64(hashtbl, "hashtbl"), _i64(cache_freq, "cache_freq")
    with _guard(dev):
        _check(lib().ttx_update_cache_state(nnz, indices.data_ptr(), hashtbl.numel(), hashtbl.data_ptr(),
                                            cache_freq.data_ptr(), _stream(dev)))


def cache_populate(num_embeddings: int, tt_p_shapes, tt_q_shapes, tt_ranks, tt_cores, L, hashtbl, cache_freq,
                   cache_state, cache_weight, reference_exact: bool = False) -> None:
    """tt_embeddings.cpp:76-86.  `reference_exact` (trailing keyword, not in the reference): leave the cache_state of evicted
    slots untouched, as the reference's mark_popular_colidx_kernel does (include/ttx.h TTX_POPULATE_REFERENCE_EXACT)."""
    g = _geom(tt_cores[0].size(0), tt_p_shapes, tt_q_shapes, tt_ranks)
    dev = _dev(cache_weight)
    cores = _cores(list(tt_cores), g)
    H = hashtbl.numel()
    if H <= 0 or H != cache_freq.numel() or H < cache_weight.size(0):  # cu:1271-1274
        raise RuntimeError("tt_embeddings: need 0 < hashtbl.numel() == cache_freq.numel() >= cache_size")
    if cache_state.dtype != torch.int32 or cache_state.numel() != H:
        raise RuntimeError("tt_embeddings: cache_state must be int32[hashtbl_size]")
    cw = cache_weight.detach()
    if cw.dtype != torch.float32 or not cw.is_contiguous():
        raise RuntimeError("tt_embeddings: cache_weight must be contiguous float32")
    cs, D = cw.size(0), cw.size(1)
    lb = lib()
    st = _stream(dev)
    nb = lb.ttx_cache_populate_workspace_bytes(C.byref(g), H, cs, D)
    ws = _workspace(dev, st, nb)
    with _guard(dev):
        _check(lb.ttx_cache_populate_f(C.byref(g), _ptr_array(cores), H, hashtbl.data_ptr(), cache_freq.data_ptr(),
                                       cache_state.data_ptr(), cs, D, cw.data_ptr(), 1 if reference_exact else 0,
                                       ws.data_ptr(), ws.numel(), st))


def preprocess_indices_sync(colidx: torch.Tensor, offsets: torch.Tensor, num_tables: int, warmup: bool,
                            hashtbl: torch.Tensor, cache_state: torch.Tensor,
                            update_cache_freq: Optional[torch.Tensor] = None
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int, Optional[torch.Tensor]]:
    """tt_embeddings.cpp:88-95.  Host-synchronous iff not warmup and num_tables == 1.
    Extra (not in the reference): `update_cache_freq` folds update_cache_state(colidx, hashtbl,
    update_cache_freq) into the same launch."""
    dev = _dev(colidx)
    colidx, offsets = _i64(colidx, "colidx"), _i64(offsets, "offsets")
    nnz = colidx.numel()
    rowidx = torch.empty_like(colidx)
    tableidx = torch.empty_like(colidx)
    if nnz == 0:
        return colidx, rowidx, tableidx, 0, None
    live = (not warmup) and num_tables == 1
    pcol = prow = ploc = None
    lb = lib()
    st = _stream(dev)
    wsp, wsn = None, 0
    if live:
        pcol, prow = torch.empty_like(colidx), torch.empty_like(colidx)
        ploc = torch.empty(nnz, dtype=torch.int32, device=dev)
        ws = _workspace(dev, st, lb.ttx_preprocess_workspace_bytes(nnz))
        wsp, wsn = ws.data_ptr(), ws.numel()
    num_tt, part = C.c_int32(0), C.c_int32(0)
    fuse = update_cache_freq is not None and hashtbl.numel() > 0
    if fuse and hashtbl.numel() != update_cache_freq.numel():
        raise RuntimeError("tt_embeddings: hashtbl must match cache_freq")
    with _guard(dev):
        _check(lb.ttx_preprocess_indices_sync_fused(
            nnz, colidx.data_ptr(), offsets.numel() - 1, offsets.data_ptr(), num_tables, int(bool(warmup)),
            hashtbl.numel(), hashtbl.data_ptr() if (live or fuse) else None, cache_state.data_ptr() if live else None,
            rowidx.data_ptr(), tableidx.data_ptr(), pcol.data_ptr() if live else None,
            prow.data_ptr() if live else None, ploc.data_ptr() if live else None,
            C.byref(num_tt), C.byref(part), hashtbl.data_ptr() if fuse else None,
            update_cache_freq.data_ptr() if fuse else None, wsp, wsn, st))
    if part.value:
        return pcol, prow, tableidx, int(num_tt.value), ploc
    return colidx, rowidx, tableidx, nnz, None


def _check_cached_args(nnz: int, cache_locations: torch.Tensor, rowidx: torch.Tensor) -> None:
    if cache_locations.dtype != torch.int32 or not cache_locations.is_contiguous():
        raise RuntimeError("tt_embeddings: cache_locations must be contiguous int32")
    if nnz > cache_locations.numel() or nnz > rowidx.numel():
        raise RuntimeError("tt_embeddings: nnz exceeds cache_locations / rowidx")


def cache_forward(B: int, nnz: int, cache_locations: torch.Tensor, rowidx: torch.Tensor, cache_weight: torch.Tensor,
                  output: torch.Tensor) -> None:
    """tt_embeddings.cpp:97-103: output[rowidx[n], :] += cache_weight[cache_locations[n], :]."""
    dev = _dev(output)
    if B <= 0:
        raise RuntimeError("tt_embeddings: B must be > 0")  # cu:1549
    if nnz == 0:
        return
    _check_cached_args(nnz, cache_locations, rowidx)
    cw = cache_weight.detach()
    with _guard(dev):
        _check(lib().ttx_cache_forward(B, nnz, cache_locations.data_ptr(), _i64(rowidx, "rowidx").data_ptr(), cw.size(1),
                                       cw.data_ptr(), output.data_ptr(), _stream(dev)))


# The cache rows' update WITHOUT atomics (ttx_cache_backward_sorted: the cached lookups grouped by cache row with a stable sort, a
# row's bag gradients added in index order, one writer per row -- bit-identical from run to run, and the sequential oracle's order
# for row-wise Adagrad).  `deterministic` (trailing keyword of the three functions below, not in the reference): True = always,
# False = the one-launch float-atomic kernels (the reference's own formulation; the last bits of cache_weight then depend on the
# order the adds arrive in), None = "auto": sorted from DETERMINISTIC_AUTO_MIN_NNZ cached lookups on (row-wise Adagrad:
# DETERMINISTIC_AUTO_MIN_NNZ_ADAGRAD), where it is also the faster of the two.  TTX_DETERMINISTIC=1 / 0 in the environment overrides None.
# (measured, profiles/r06_cache_bandwidth.md: the sorted update is a chain of ~16 launches, ~65 us whatever the batch; on a Zipf
#  stream it overtakes the atomic SGD / dense scatter near 300k cached lookups -- 232 against 343 us at 1 M -- and the atomic
#  row-wise Adagrad, whose hot rows serialise, near 60k -- 357 against 1228 us at 1 M)
DETERMINISTIC_AUTO_MIN_NNZ = int(os.environ.get("TTX_DETERMINISTIC_AUTO_MIN_NNZ", 262144))
DETERMINISTIC_AUTO_MIN_NNZ_ADAGRAD = int(os.environ.get("TTX_DETERMINISTIC_AUTO_MIN_NNZ_ADAGRAD", 65536))


def _use_sorted(deterministic: Optional[bool], nnz: int, adagrad: bool = False) -> bool:
    if deterministic is None and os.environ.get("TTX_DETERMINISTIC", "") != "":
        deterministic = os.environ["TTX_DETERMINISTIC"] not in ("0", "")
    if deterministic is None:
        return nnz >= (DETERMINISTIC_AUTO_MIN_NNZ_ADAGRAD if adagrad else DETERMINISTIC_AUTO_MIN_NNZ)
    return bool(deterministic)


def _cache_backward_sorted(optim: int, nnz: int, go: torch.Tensor, cache_locations: torch.Tensor, rowidx: torch.Tensor, lr: float,
                           eps: float, state: Optional[torch.Tensor], dst: torch.Tensor, skip_dev: Optional[torch.Tensor] = None) -> None:
    dev = _dev(dst)
    D, cs = dst.size(1), dst.size(0)
    num_bags = go.numel() // D
    lb = lib()
    st = _stream(dev)
    nb = lb.ttx_cache_backward_sorted_workspace_bytes(nnz, num_bags, D)
    ws = _workspace(dev, st, nb)
    with _guard(dev):
        _check(lb.ttx_cache_backward_sorted(optim, nnz, None if skip_dev is None else skip_dev.data_ptr(), num_bags, D, go.data_ptr(),
                                            cache_locations.data_ptr() if nnz else None,
                                            _i64(rowidx, "rowidx").data_ptr() if nnz else None, lr, eps, cs,
                                            None if state is None else state.data_ptr(), dst.data_ptr(),
                                            ws.data_ptr() if nnz else None, ws.numel(), st))


def cache_backward_sgd(nnz: int, grad_output: torch.Tensor, cache_locations: torch.Tensor, rowidx: torch.Tensor,
                       learning_rate: float, cache_weight: torch.Tensor, deterministic: Optional[bool] = None) -> None:
    """tt_embeddings.cpp:105-111."""
    if nnz == 0:
        return
    dev = _dev(cache_weight)
    cw = cache_weight.detach()
    go = _f32(grad_output, "grad_output")
    _check_cached_args(nnz, cache_locations, rowidx)
    if _use_sorted(deterministic, nnz):
        return _cache_backward_sorted(OPTIM_SGD, nnz, go, cache_locations, rowidx, learning_rate, 0.0, None, cw)
    with _guard(dev):
        _check(lib().ttx_cache_backward_sgd(nnz, cw.size(1), go.data_ptr(), cache_locations.data_ptr(),
                                            _i64(rowidx, "rowidx").data_ptr(), learning_rate, cw.data_ptr(), _stream(dev)))


def cache_backward_dense(nnz: int, grad_output: torch.Tensor, cache_locations: torch.Tensor, rowidx: torch.Tensor,
                         learning_rate: float, cache_weight: torch.Tensor, deterministic: Optional[bool] = None) -> torch.Tensor:
    """tt_embeddings.cpp:113-119 (learning_rate unused, as in the reference)."""
    dev = _dev(cache_weight)
    cw = cache_weight.detach()
    out = torch.empty_like(cw)
    go = _f32(grad_output, "grad_output")
    _check_cached_args(nnz, cache_locations, rowidx)
    if _use_sorted(deterministic, nnz):
        _cache_backward_sorted(OPTIM_DENSE, nnz, go, cache_locations, rowidx, 0.0, 0.0, None, out)
        return out
    with _guard(dev):
        _check(lib().ttx_cache_backward_dense(nnz, cw.size(1), go.data_ptr(),
                                              cache_locations.data_ptr() if nnz else None,
                                              _i64(rowidx, "rowidx").data_ptr() if nnz else None, cw.size(0),
                                              out.data_ptr(), _stream(dev)))
    return out


def cache_backward_rowwise_adagrad_approx(nnz: int, grad_output: torch.Tensor, cache_locations: torch.Tensor,
                                          rowidx: torch.Tensor, learning_rate: float, eps: float,
                                          cache_optimizer_state: torch.Tensor, cache_weight: torch.Tensor,
                                          deterministic: Optional[bool] = None) -> None:
    """tt_embeddings.cpp:121-129."""
    if nnz == 0:
        return
    dev = _dev(cache_weight)
    cw = cache_weight.detach()
    go = _f32(grad_output, "grad_output")
    _dev(cache_optimizer_state)
    _check_cached_args(nnz, cache_locations, rowidx)
    if _use_sorted(deterministic, nnz, adagrad=True):
        return _cache_backward_sorted(OPTIM_ADAGRAD, nnz, go, cache_locations, rowidx, learning_rate, eps, cache_optimizer_state, cw)
    with _guard(dev):
        _check(lib().ttx_cache_backward_rowwise_adagrad_approx(
            nnz, cw.size(1), go.data_ptr(), cache_locations.data_ptr(), _i64(rowidx, "rowidx").data_ptr(),
            learning_rate, eps, cache_optimizer_state.data_ptr(), cw.data_ptr(), _stream(dev)))


# --------------------------------------------------------------------------- #
# extras
# --------------------------------------------------------------------------- #

def tt_rows(num_tables, D, tt_p_shapes, tt_q_shapes, tt_ranks, indices, tableidx, tt_cores) -> torch.Tensor:
    """Decompress rows: rows[n, :] = TT row of indices[n] (contraction only)."""
    g = _geom(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks)
    dev = _dev(tt_cores[0])
    cores = _cores(tt_cores, g)
    indices = _i64(indices, "indices")
    nnz = indices.numel()
    rows = torch.empty((nnz, D), dtype=torch.float32, device=dev)
    lb = lib()
    st = _stream(dev)
    ws = _workspace(dev, st, lb.ttx_plan_bytes(C.byref(g), nnz) + 256)
    with _guard(dev):
        _check(lb.ttx_tt_rows(C.byref(g), D, nnz, indices.data_ptr(),
                              None if tableidx is None else _i64(tableidx, "tableidx").data_ptr(), _ptr_array(cores),
                              rows.data_ptr(), ws.data_ptr(), ws.numel(), st))
    return rows


def profile_enable(mask: int) -> None:
    """bit w of `mask` turns on live HIP-event timing of kernel slot w (PROF_*); 0 = off."""
    _check(lib().ttx_profile_enable(int(mask)))


def profile_mask(mask: int) -> None:
    """profile_enable without draining: use around a hipGraph capture, read after the replays."""
    _check(lib().ttx_profile_mask(int(mask)))


def profile_reset() -> None:
    _check(lib().ttx_profile_reset())


def profile_read(which: int) -> Tuple[int, float]:
    n, ms = C.c_int64(0), C.c_double(0.0)
    _check(lib().ttx_profile_read(which, C.byref(n), C.byref(ms)))
    return int(n.value), float(ms.value)


def debug_sort_pairs_desc(keys: torch.Tensor, vals: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """test hook: cache_populate's stable descending 64-bit radix sort of (key, value) pairs on its own"""
    keys, vals = _i64(keys, "keys"), _i64(vals, "vals")
    dev = _dev(keys)
    n = keys.numel()
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    lb = lib()
    lb.ttx_debug_sort_workspace_bytes.restype = C.c_size_t
    lb.ttx_debug_sort_workspace_bytes.argtypes = [C.c_int64]
    lb.ttx_debug_sort_pairs_desc.argtypes = [C.c_int64] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    nb = lb.ttx_debug_sort_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    with _guard(dev):
        _check(lb.ttx_debug_sort_pairs_desc(n, keys.data_ptr(), vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), ws.data_ptr(), nb,
                                            _stream(dev)))
    return ko, vo


# ---- test / ablation knobs: libttx_hooks.so (the product library has none) ----
def set_chunk(mc: int) -> None:
    """indices per work-group chunk (0 = heuristic)"""
    _check(hooks_lib().ttx_set_chunk(int(mc)))
    _knobs_changed()


def debug_bwd32(lookups_per_chunk: int) -> None:
    """experiment (round 6): the backward of q = [4,4,4], ranks [32,32] at >= 131072 lookups on bwd32_kernel (v_mfma_f32_32x32x2,
    persistent work-groups) with this many lookups per chunk (a multiple of 32); 0 = spec_bwd_kernel, the product's kernel"""
    _check(hooks_lib().ttx_debug_bwd32(int(lookups_per_chunk)))
    _knobs_changed()


def debug_lds_budget(nbytes: int) -> None:
    """LDS budget of the generic kernels' tile search (0 = the hardware's 160 KiB)."""
    _check(hooks_lib().ttx_debug_lds_budget(int(nbytes)))
    _knobs_changed()


def debug_skip(mask: int) -> None:
    """bit 8: force the generic kernels; bit 15: exact-shape templates only; bit 16: reduce_apply with a work-group (not a wave) per
    small slice -- these three leave the results valid; the other bits skip kernel phases / launches
    (results INVALID).  0 = normal operation."""
    _check(hooks_lib().ttx_debug_skip(int(mask)))
    _knobs_changed()


def debug_cache_fwd(lookup_groups: int) -> None:
    _check(hooks_lib().ttx_debug_cache_fwd(int(lookup_groups)))
    _knobs_changed()


def debug_stamps(ptr: Optional[int]) -> None:
    _check(hooks_lib().ttx_debug_stamps(C.c_void_p(ptr or None)))
    _knobs_changed()


def debug_tiles(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks) -> dict:
    """Test helper: the block walk the generic kernels take for this geometry under the current budget."""
    g = _geom(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks)
    out = (C.c_int32 * 6)()
    _check(lib().ttx_debug_tiles(C.byref(g), out))
    return dict(zip(("MC", "bpp", "KB", "ncp", "nkb", "bytes"), [int(x) for x in out]))
