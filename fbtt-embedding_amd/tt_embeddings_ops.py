"""`tt_embeddings_ops` -- the TTEmbeddingBag module surface on top of libttx.

Same public names, constructor keywords, forward() call form, autograd contract
and state_dict keys as the reference's tt_embeddings_ops.py, so a model that
uses `TTEmbeddingBag` / `TableBatchedTTEmbeddingBag` in place of
`nn.EmbeddingBag(mode="sum", include_last_offset=True)` switches by import
path only.  All compute goes through the module `tt_embeddings` of this package
(ctypes -> C ABI -> hand-written HIP for gfx950); nothing here computes on the
CPU.

Reference map (file:line in /root/reference/tt_embeddings_ops.py):
  OptimType :18-33 | BufferList :36-77 | tt_matrix_to_full :80-127 |
  TTLookupFunction :130-356 | suggested_tt_shapes :359-418 |
  TableBatchedTTEmbeddingBag :421-886 | TTEmbeddingBag :889-934

Deliberate differences (each is a superset or a fix):
  * optional trailing ctor keyword `device` (default: current GPU);
  * D % 4 != 0 is supported; the README's toy example (E=10, D=3) runs;
  * `cache_optimizer_state` lives on the module's device (the reference leaves
    it on the CPU, :582-585); `reset_cache()` works (:795 has a typo);
    `get_params()` does not grow the ParameterList on every call (:882-886);
  * fused optimizers update every looked-up slice (the reference's apply-kernel
    grid skips rows);
  * nn.EmbeddingBag call forms: ctor keyword `include_last_offset` (default True =
    the reference's form), int32 indices / offsets, forward keyword
    `per_sample_weights`;
  * when ttx_torch.so is built the lookup runs as a C++ autograd node (same C ABI
    calls as TTLookupFunction below, which stays the reference-shaped route); with a
    live cache that node keeps the partition's split point on the device instead of
    reading it back every step, and in dense mode it always returns a (possibly zero)
    gradient for cache_weight.
"""
import itertools
import logging
import math
import os
from operator import is_ as _is
from weakref import ref as _weakref
import random
from enum import Enum, unique
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

import tt_embeddings as _engine  # the 11-function native-module surface

_native = None  # ttx_torch (csrc/ttx_torch.cpp), False once an import attempt failed


def _native_node():
    """The C++ autograd node of the lookup, when it was built (__graft_entry__.build()) and the engine is
    the HIP shim (tests swap `_engine` for the oracle-backed stand-in on CPU).  It is an optional,
    faster route through the SAME C ABI -- TTLookupFunction below stays the reference-shaped path."""
    global _native
    if getattr(_engine, "__name__", "") != "tt_embeddings" or os.environ.get("TTX_NO_NATIVE_NODE"):
        return None
    if _engine._lib is not None and _engine._lib is _engine._hooks:
        return None  # a test / ablation knob is set: calls go to libttx_hooks.so, which the C++ node is not linked against
    if _native is None:
        try:
            import ttx_torch as _m

            _native = _m
        except ImportError:
            _native = False
    return _native or None


# ---- `out.backward(grad)` of a lookup's own output, past autograd's engine: OPT-IN --------------------------------------------------
# The reference benchmark's loop is `tt_emb(indices, offsets).backward(grad)` (tt_embeddings_benchmark.py:94-108); with a fused
# optimizer the graph under that output is the lookup's one node, which returns no gradient to anybody, and autograd's engine
# spends ~40 us of host time per step (graph task, hand-over to its device thread and back) on calling it -- more than the step's
# kernels take.  `enable_direct_backward()` (or TTX_DIRECT_BACKWARD=1 in the environment at import) wraps `torch.Tensor.backward`
# ONCE; importing this module leaves torch untouched (round 6: a drop-in that rewrites a core torch method on import is not
# reviewable; in a DLRM the lookup's output is never the tensor `.backward()` is called on, so this serves benchmark-shaped loops
# only).  While enabled, the module registers the tensors it returns on the C++ node's route (`_direct_register`), and the wrapper
# calls the node on the calling thread for exactly those tensors (csrc/ttx_torch.cpp NodeRef::backward_of) when the call is the
# plain one: a gradient and nothing else (no retain_graph / create_graph / inputs=), a plain torch.Tensor (no subclass), no
# TorchFunctionMode active, no hooks or retain_grad() on the tensor, and the tensor's grad_fn STILL the lookup's node (an in-place
# op on the output rebases it: the engine's job); NodeRef::backward itself declines on hooks on the node, anomaly mode, another
# current stream than the forward's, a gradient that is part of a graph or has the wrong shape.  Everything declined, every other
# tensor and every other USE of the output (operand of further ops, torch.autograd.backward / grad) is the original
# Tensor.backward.  The registry is a side table keyed by id() with a weak reference per entry: nothing is stored ON the tensor
# (torch.save / pickle of an output see no foreign attribute), an entry goes when its tensor goes.
# `disable_direct_backward()` puts the original method back.
_DIRECT_BACKWARD = False
_direct: Dict[int, tuple] = {}  # id(output tensor) -> (weak reference to it, NodeRef)
_tensor_backward = None  # torch.Tensor.backward as it was when enable_direct_backward() wrapped it


class _KeyRef(_weakref):
    __slots__ = ("key",)


def _direct_drop(wr) -> None:
    _direct.pop(wr.key, None)


def _direct_register(out: torch.Tensor, ref) -> None:
    wr = _KeyRef(out, _direct_drop)
    wr.key = id(out)
    _direct[wr.key] = (wr, ref)


def _backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
    if _direct:
        e = _direct.get(id(self))
        # (a gradient of another type or shape goes to the engine for the engine's own error)
        if (e is not None and e[0]() is self and _DIRECT_BACKWARD and type(self) is torch.Tensor and type(gradient) is torch.Tensor
                and not retain_graph and not create_graph and inputs is None and gradient.shape == self.shape
                and not self._backward_hooks and not self.retains_grad and not torch._C._len_torch_function_stack()
                and e[1].backward_of(self, gradient)):
            return None
    return _tensor_backward(self, gradient, retain_graph, create_graph, inputs)


def enable_direct_backward() -> None:
    """Opt in: `out.backward(grad)` of a fused-optimizer lookup's own output calls the lookup's node on the calling thread
    (see above).  Wraps torch.Tensor.backward once, process-wide, until disable_direct_backward()."""
    global _DIRECT_BACKWARD, _tensor_backward
    if torch.Tensor.backward is not _backward:
        _tensor_backward = torch.Tensor.backward
        _backward.__doc__ = _tensor_backward.__doc__
        torch.Tensor.backward = _backward
    _DIRECT_BACKWARD = True


def disable_direct_backward() -> None:
    """Undo enable_direct_backward(): torch.Tensor.backward is the method it was; registered outputs are forgotten."""
    global _DIRECT_BACKWARD
    _DIRECT_BACKWARD = False
    _direct.clear()
    if torch.Tensor.backward is _backward:
        torch.Tensor.backward = _tensor_backward


def direct_backward_enabled() -> bool:
    return _DIRECT_BACKWARD and torch.Tensor.backward is _backward


if os.environ.get("TTX_DIRECT_BACKWARD", "0") not in ("", "0"):
    enable_direct_backward()


@unique
class OptimType(Enum):
    SGD = "sgd"
    EXACT_SGD = "exact_sgd"
    LAMB = "lamb"
    ADAM = "adam"
    EXACT_ADAGRAD = "exact_adagrad"
    EXACT_ROWWISE_ADAGRAD = "exact_row_wise_adagrad"
    LARS_SGD = "lars_sgd"
    PARTIAL_ROWWISE_ADAM = "partial_row_wise_adam"
    PARTIAL_ROWWISE_LAMB = "partial_row_wise_lamb"

    def __str__(self) -> str:
        return self.value


_SGD_LIKE = (OptimType.SGD, OptimType.EXACT_SGD)  # everything else -> Adagrad kernel (:221,:248)
# dedup="auto" policy (measured crossover): the map + pre-sum + gather pooling cost about a quarter of the plain
# step at 327k lookups, so sharing pays below ~0.7 distinct pairs per lookup; 0.6 leaves a margin
_DEDUP_AUTO_MIN_NNZ = int(os.environ.get("TTX_DEDUP_AUTO_MIN_NNZ", 65536))
_DEDUP_AUTO_MAX_DISTINCT = 0.6
_DEDUP_AUTO_PERIOD = 256


def _split0_factor(q: Sequence[int], ranks: Sequence[int]) -> int:
    """k > 1: a T = 3 table with q0 = k q0' (2 <= q0' <= 4, k <= 4), q1, q2 <= 8 and ranks <= 128 is contracted as k part
    lookups per index in the table [k p0, p1, p2] x [q0', q1, q2] -- core 0 [p0, q0, r1] IS [k p0, q0', r1] -- which the
    shape-specialised kernels take (include/ttx.h "core-0 row split"; the reference's default factoring of D = 512 is
    [8, 8, 8]).  0: no split."""
    if len(q) != 3 or q[0] <= 4 or q[1] > 16 or q[2] > 32 or max(ranks) > 128 or os.environ.get("TTX_NO_SPLIT0"):
        return 0
    if q[2] > 16 and max(ranks) > 32:  # (q2 up to 32 -- a prime last factor 17 .. 31, round 5 -- at ranks <= 32)
        return 0
    if q[2] > 8 and max(ranks) > 64:  # (q2 up to 16 -- D = 640 / 768 / 1024 -- has templates at ranks <= 64 only)
        return 0
    if q[1] > 8 and max(ranks) > 64:  # (q1 up to 16 -- D = 720 / 800 / 864 / 880 / 960 / 1008, round 5 -- at ranks <= 64)
        return 0
    for k in (2, 3, 4):
        if q[0] % k == 0 and 2 <= q[0] // k <= 4:
            return k
    return 0


def _pad0_target(q: Sequence[int], ranks: Sequence[int]) -> int:
    """Q > q0: a T = 3 table whose q0 > 4 has no exact part split (5, 7, 10, 11, 13, 14, 15) is contracted with core 0 stored
    zero-padded to Q = the next multiple of 4 slots -- [p0, Q, r1] IS [Q/4 p0, 4, r1], the part lookups of `_split0_factor` --
    and the output rows' first D values kept (q0 is the outermost factor of an output row: the real values are a prefix of
    the padded row).  Exact: the padded slots of core 0 are zero and stay zero (their gradient has the zero output gradient
    as a factor); costs the multiply-adds on them.  The reference's default factorings of D = 320 / 448 are [5, 8, 8] /
    [7, 8, 8].  0: no padding."""
    if len(q) != 3 or q[0] <= 4 or q[0] > 16 or _split0_factor(q, ranks):
        return 0
    Q = -(-q[0] // 4) * 4
    return Q if _split0_factor([Q] + list(q[1:]), ranks) else 0


class BufferList(nn.Module):
    """An indexable list of registered buffers named `<name><i>` (state_dict
    keys `optimizer_state.optimizer_state0`, ...)."""

    def __init__(self, name: str, buffers: Optional[Sequence[torch.Tensor]] = None) -> None:
        super().__init__()
        self._name = name
        self._count = 0
        for b in buffers or ():
            self.append(b)

    def append(self, buffer: torch.Tensor) -> "BufferList":
        self.register_buffer(f"{self._name}{self._count}", buffer)
        self._count += 1
        return self

    def extend(self, buffers: Sequence[torch.Tensor]) -> "BufferList":
        for b in buffers:
            self.append(b)
        return self

    def __len__(self) -> int:
        return self._count

    def __getitem__(self, index: int) -> torch.Tensor:
        if not -self._count <= index < self._count:
            raise IndexError(index)
        return getattr(self, f"{self._name}{index % self._count}")

    def __iter__(self) -> Iterator[torch.Tensor]:
        return (self[i] for i in range(self._count))


def tt_matrix_to_full(tt_p_shapes: List[int], tt_q_shapes: List[int], tt_ranks: List[int],
                      tt_cores: Sequence[torch.Tensor], tt_permute: Optional[List[int]] = None) -> torch.Tensor:
    """Expand TT cores to the dense [prod(p), prod(q)] matrix (test oracle /
    `full_weight`).  With tt_permute=[1,0,2,3] the cores are in the module's
    storage layout [1, p, r*q*r']; otherwise they are [r, p, q, r'] tensors."""
    T = len(tt_p_shapes)
    ranks = list(tt_ranks)
    if len(ranks) == T - 1:
        ranks = [1] + ranks + [1]
    mats = []
    for t, core in enumerate(tt_cores):
        shape = [ranks[t], tt_p_shapes[t], tt_q_shapes[t], ranks[t + 1]]
        if tt_permute is not None:
            stored = [shape[a] for a in tt_permute]
            core = core.reshape(stored).permute(*tt_permute)
        else:
            core = torch.squeeze(core)
        if list(core.shape) != shape:
            raise ValueError(f"core {t}: expected {shape}, got {list(core.shape)}")
        mats.append(core.contiguous())
    acc = mats[0]
    for t in range(1, T):
        acc = acc.reshape(-1, ranks[t]) @ mats[t].reshape(ranks[t], -1)
    inter = [d for pq in zip(tt_p_shapes, tt_q_shapes) for d in pq]
    acc = acc.reshape(inter)
    order = list(range(0, 2 * T, 2)) + list(range(1, 2 * T, 2))
    n_rows = int(np.prod(np.asarray(tt_p_shapes, dtype=np.int64)))
    n_cols = int(np.prod(np.asarray(tt_q_shapes, dtype=np.int64)))
    return acc.permute(order).contiguous().reshape(n_rows, n_cols).float()


class TTLookupFunction(torch.autograd.Function):
    """Autograd node of one (table-batched) TT lookup.  Argument order and the
    gradient tuple (19 x None, `d_cache_weight` in slot 17 for dense mode, then
    one entry per core) follow the reference (:133-155, :280-356)."""

    @staticmethod
    def forward(ctx, B: int, D: int, tt_p_shapes: List[int], tt_q_shapes: List[int], tt_ranks: List[int],
                L: torch.Tensor, nnz_tt: int, nnz_cached: int, indices: torch.Tensor, rowidx: torch.Tensor,
                tableidx: torch.Tensor, optimizer: OptimType, learning_rate: float, eps: float, sparse: bool,
                cache_locations: Optional[torch.Tensor], cache_optimizer_state: Optional[torch.Tensor],
                cache_weight: Optional[torch.Tensor], optimizer_state: List[torch.Tensor],
                *tt_cores: torch.Tensor) -> torch.Tensor:
        ctx.geometry = (tt_p_shapes, tt_q_shapes, tt_ranks)
        ctx.D = D
        ctx.optimizer, ctx.learning_rate, ctx.eps, ctx.sparse = optimizer, learning_rate, eps, sparse
        ctx.tt_cores = tt_cores
        ctx.optimizer_state = optimizer_state
        ctx.nnz_tt, ctx.nnz_cached = nnz_tt, nnz_cached
        ctx.has_cache = cache_weight is not None
        ctx.save_for_backward(L, indices, rowidx, tableidx, cache_locations, cache_optimizer_state, cache_weight)
        num_tables = tt_cores[0].size(0)
        if len(tt_p_shapes) > 0 and isinstance(tt_p_shapes[0], (list, tuple)):
            num_tables = len(tt_p_shapes)  # tables of different row factors: cores are [1, sum p, slice]
        # one lookup plan serves forward and backward of this batch
        mk = getattr(_engine, "make_plan", None)
        ctx.plan = getattr(rowidx, "_ttx_plan", None)  # built by the module's lookup prologue
        ctx.det = getattr(rowidx, "_ttx_det", None)    # the module's deterministic_cache_update (None: the engine's default)
        if ctx.plan is None and mk is not None:
            ctx.plan = mk(num_tables, tt_p_shapes, tt_q_shapes, tt_ranks, nnz_tt, indices, tableidx, rowidx)
        extra = {"plan": ctx.plan} if ctx.plan is not None else {}
        output = _engine.tt_forward(1000, num_tables, B, D, tt_p_shapes, tt_q_shapes, tt_ranks, L, nnz_tt, indices,
                                    rowidx, tableidx, list(tt_cores), **extra)
        if nnz_cached > 0:
            _engine.cache_forward(B, nnz_cached, cache_locations[nnz_tt:], rowidx[nnz_tt:], cache_weight, output)
        return output

    @staticmethod
    def backward(ctx, d_output: torch.Tensor):
        L, indices, rowidx, tableidx, cache_locations, cache_optimizer_state, cache_weight = ctx.saved_tensors
        p, q, ranks = ctx.geometry
        n_tt, n_c = ctx.nnz_tt, ctx.nnz_cached
        extra = {"plan": ctx.plan} if ctx.plan is not None else {}
        det = {"deterministic": ctx.det} if ctx.det is not None else {}
        cores = list(ctx.tt_cores)
        d_output = d_output.contiguous()
        head: List[Optional[torch.Tensor]] = [None] * 19
        if ctx.sparse:
            if ctx.optimizer in _SGD_LIKE:
                _engine.tt_sgd_backward(1000, ctx.D, ctx.learning_rate, p, q, ranks, L, n_tt, indices, rowidx, tableidx,
                                        d_output, cores, **extra)
                if n_c > 0:
                    _engine.cache_backward_sgd(n_c, d_output, cache_locations[n_tt:], rowidx[n_tt:],
                                               ctx.learning_rate, cache_weight, **det)
            else:
                _engine.tt_adagrad_backward(1000, ctx.D, ctx.learning_rate, ctx.eps, p, q, ranks, L, n_tt, indices,
                                            rowidx, tableidx, d_output, ctx.optimizer_state, cores, **extra)
                if n_c > 0:
                    _engine.cache_backward_rowwise_adagrad_approx(n_c, d_output, cache_locations[n_tt:],
                                                                  rowidx[n_tt:], ctx.learning_rate, ctx.eps,
                                                                  cache_optimizer_state, cache_weight, **det)
            return tuple(head + [None] * len(cores))
        grads = _engine.tt_dense_backward(1000, ctx.D, p, q, ranks, L, n_tt, indices, rowidx, tableidx, d_output,
                                          cores, **extra)
        if n_c > 0:
            head[17] = _engine.cache_backward_dense(n_c, d_output, cache_locations[n_tt:], rowidx[n_tt:],
                                                    ctx.learning_rate, cache_weight, **det)
        return tuple(head + list(grads))


# --------------------------------------------------------------------------- #
# shape factoring (init-time helper; reference :359-418)
# --------------------------------------------------------------------------- #

def _prime_factors(n: int) -> Dict[int, int]:
    out: Dict[int, int] = {}
    f = 2
    while f * f <= n:
        while n % f == 0:
            out[f] = out.get(f, 0) + 1
            n //= f
        f += 1 if f == 2 else 2
    if n > 1:
        out[n] = out.get(n, 0) + 1
    return out


def _entropy(xs: Sequence[int]) -> float:
    v = np.asarray(xs, dtype=np.float64)
    pk = v / v.sum()
    pk = pk[pk > 0]
    return float(-(pk * np.log(pk)).sum())


def _balanced_factors(n: int, d: int) -> List[int]:
    """The factorisation of n into d factors with maximal entropy of the
    normalised factors, listed small/large interleaved like the reference."""
    primes = _prime_factors(n)
    splits = []  # per prime: all ways to spread its exponent over d ordered bins
    for prime, e in primes.items():
        ways = []
        for cuts in itertools.combinations(range(e + d - 1), d - 1):
            prev, exps = -1, []
            for c in cuts + (e + d - 1,):
                exps.append(c - prev - 1)
                prev = c
            ways.append([prime ** k for k in exps])
        splits.append(ways)
    seen = set()
    for combo in itertools.product(*splits) if splits else [()]:
        bins = [1] * d
        for way in combo:
            bins = [b * w for b, w in zip(bins, way)]
        seen.add(tuple(sorted(bins)))
    best = max(sorted(seen), key=_entropy)
    lo, hi = list(best[: d // 2]), list(best[d // 2:])
    out: List[int] = []
    for a, b in itertools.zip_longest(lo, hi):
        if a is not None:
            out.append(a)
        if b is not None:
            out.append(b)
    return out


def suggested_tt_shapes(n: int, d: int = 3, allow_round_up: bool = True) -> List[int]:
    n = int(n)
    if not allow_round_up:
        return _balanced_factors(n, d)
    candidates = []
    for k in range(len(str(n))):
        step = 10 ** k
        candidates.append(_balanced_factors(-(-n // step) * step, d))
    return candidates[int(np.argmax([_entropy(c) for c in candidates]))]


# --------------------------------------------------------------------------- #
# modules
# --------------------------------------------------------------------------- #

class TableBatchedTTEmbeddingBag(nn.Module):
    """`num_tables` TT-compressed embedding tables of identical shape looked up
    in one pass (sum pooling, include_last_offset form: offsets has
    num_tables*B + 1 entries, bags ordered table-major)."""

    __constants__ = ["num_tables", "num_embeddings", "embedding_dim", "tt_shape", "tt_rank"]

    def __init__(self, num_tables: int, num_embeddings: int, embedding_dim: int, tt_ranks: List[int],
                 tt_p_shapes: Optional[List[int]] = None, tt_q_shapes: Optional[List[int]] = None,
                 optimizer: OptimType = OptimType.SGD, learning_rate: float = 0.1, eps: float = 1.0e-10,
                 sparse: bool = True, use_cache: bool = False, cache_size: int = 0, hashtbl_size: int = 0,
                 weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False,
                 device: Optional[torch.device] = None, include_last_offset: bool = True, dedup: bool = False,
                 reference_exact_populate: bool = False, deterministic_cache_update: Optional[bool] = None) -> None:
        super().__init__()
        # deterministic_cache_update (trailing keyword, not in the reference): how the backward updates the CACHE rows.  True: the
        # cached lookups are grouped by cache row with a stable sort, a row's bag gradients added in index order, one writer per row --
        # cache_weight (and the row-wise Adagrad state) bit-identical from run to run (ttx_cache_backward_sorted).  False: the
        # reference's formulation, float atomics in one launch -- the last bits depend on arrival order.  None: sorted from 262,144
        # lookups per batch on (row-wise Adagrad: 65,536), where it is also the faster of the two.  The TT
        # cores' update never uses atomics.
        self.deterministic_cache_update = deterministic_cache_update
        # reference_exact_populate (trailing keyword, not in the reference): cache_populate() leaves the cache_state of evicted
        # slots untouched, bit for bit what the reference's mark_popular_colidx_kernel does (cu:1131-1133).  Default False: an
        # evicted slot drops its cache row (deviation 5).  Carried per call (TTX_POPULATE_REFERENCE_EXACT);
        # rounds 3-5 had a process-wide switch for it.
        self.reference_exact_populate = bool(reference_exact_populate)
        # dedup (not in the reference): lookups of a batch that repeat a (table, row) pair share ONE contraction,
        # forward and backward (include/ttx.h "duplicate lookups").  Same results; pays on skewed index streams
        # while the row cache is not live, costs the key sort and two extra launches on uniform ones.
        # dedup="auto": the module decides per batch -- sharing runs only at batches of >= _DEDUP_AUTO_MIN_NNZ lookups
        # (below that every kernel of the plain step sits at its launch floor: sharing cannot pay) and only
        # while the stream's last sampled batch had <= _DEDUP_AUTO_MAX_DISTINCT distinct pairs per lookup; the sample is
        # the shared path itself (its map holds the distinct count; one read-back every _DEDUP_AUTO_PERIOD batches, never
        # under stream capture).
        self.dedup = "auto" if dedup == "auto" else bool(dedup)
        self._dd_auto = [False, 0]  # [sharing on, batches until the next sample]
        # nn.EmbeddingBag call form: False = offsets hold only the bag starts (PyTorch's default,
        # what DLRM passes); True = the reference's form, num_tables*B + 1 entries (:851)
        self.include_last_offset = bool(include_last_offset)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("TTEmbeddingBag needs a GPU (the reference asserts torch.cuda.is_available(), :454)")
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        num_embeddings, embedding_dim = int(num_embeddings), int(embedding_dim)
        assert num_tables > 0 and num_embeddings > 0 and embedding_dim > 0
        assert num_tables == 1 or not use_cache, "cannot use cache when num_tables != 1"
        tt_ranks = [int(x) for x in tt_ranks]
        nd = len(tt_ranks) + 1
        self.tt_p_shapes: List[int] = [int(x) for x in tt_p_shapes] if tt_p_shapes is not None \
            else suggested_tt_shapes(num_embeddings, nd)
        self.tt_q_shapes: List[int] = [int(x) for x in tt_q_shapes] if tt_q_shapes is not None \
            else suggested_tt_shapes(embedding_dim, nd, allow_round_up=not enforce_embedding_dim)
        assert 2 <= len(self.tt_p_shapes) <= 4
        assert len(self.tt_p_shapes) == nd == len(self.tt_q_shapes)
        assert all(v > 0 for v in self.tt_p_shapes + self.tt_q_shapes + tt_ranks)
        assert int(np.prod(np.asarray(self.tt_p_shapes, dtype=np.int64))) >= num_embeddings
        assert int(np.prod(np.asarray(self.tt_q_shapes, dtype=np.int64))) == embedding_dim
        self.num_tables, self.tt_ndim = num_tables, nd
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.tt_ranks = [1] + tt_ranks + [1]
        self._split0 = _split0_factor(self.tt_q_shapes, tt_ranks)  # (q0 > 4: part lookups on the q0 <= 4 kernels)
        self._pad0 = _pad0_target(self.tt_q_shapes, tt_ranks)       # (... of core 0 padded to a multiple of 4 slots, round 4)
        if self._pad0:
            self._split0 = self._pad0 // 4
        if self._split0 > 1 and getattr(_engine, "debug_tiles", None) is not None and not isinstance(self.tt_p_shapes[0], (list, tuple)):
            # ... if the engine has a specialised kernel for the PART geometry (a padded template may be refused for wasting
            # more than 8x its work, the specialised kernels may be switched off): k times the lookups on the generic kernels
            # would be slower than the unsplit table on them (round 3 advisor)
            try:
                k = self._split0
                part = _engine.debug_tiles(num_tables, [k * self.tt_p_shapes[0]] + list(self.tt_p_shapes[1:]),
                                           [(self._pad0 or self.tt_q_shapes[0]) // k] + list(self.tt_q_shapes[1:]), list(self.tt_ranks))
                if part["MC"] != 0:
                    self._split0 = self._pad0 = 0
            except Exception:  # noqa: BLE001 -- the query is advice, never a reason to fail construction
                pass
        self.sparse, self.optimizer, self.learning_rate, self.eps = sparse, optimizer, learning_rate, eps
        logging.info("Creating TTEmbeddingBag tt_p_shapes: %s, tt_q_shapes: %s, tt_ranks: %s, sparse: %s, "
                     "optimizer: %s, learning_rate: %s, eps: %s, use_cache: %s, cache_size: %s, hashtbl_size: %s",
                     self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks, sparse, optimizer, learning_rate, eps,
                     use_cache, cache_size, hashtbl_size)
        strides = [int(np.prod(np.asarray(self.tt_p_shapes[t + 1:], dtype=np.int64))) for t in range(nd)]
        self.register_buffer("L", torch.tensor(strides, dtype=torch.int64, device=device))
        self.tt_cores = nn.ParameterList()
        self.optimizer_state = BufferList("optimizer_state")
        stateful = optimizer not in _SGD_LIKE
        for t in range(nd):
            shape = (num_tables, self.tt_p_shapes[t], self.tt_ranks[t] * self.tt_q_shapes[t] * self.tt_ranks[t + 1])
            self.tt_cores.append(nn.Parameter(torch.empty(shape, device=device, dtype=torch.float32)))
            self.optimizer_state.append(torch.zeros(shape if stateful else 0, device=device, dtype=torch.float32))
        self.reset_parameters(weight_dist)
        self.use_cache = use_cache
        if use_cache:
            cache_size = cache_size if cache_size > 0 else int(0.1 * num_embeddings)
            hashtbl_size = hashtbl_size if hashtbl_size > 0 else num_embeddings
            assert hashtbl_size >= cache_size
            self.register_buffer("hashtbl", torch.full((hashtbl_size,), -1, device=device, dtype=torch.int64))
            self.register_buffer("cache_freq", torch.zeros(hashtbl_size, device=device, dtype=torch.int64))
            self.register_buffer("cache_state", torch.full((hashtbl_size,), -1, device=device, dtype=torch.int32))
            self.cache_weight = nn.Parameter(torch.zeros((cache_size, embedding_dim), device=device, dtype=torch.float32))
            if sparse and stateful:
                shape = (cache_size, embedding_dim) if optimizer == OptimType.EXACT_ADAGRAD else (cache_size,)
                self.register_buffer("cache_optimizer_state", torch.zeros(shape, device=device, dtype=torch.float32))
            else:
                self.cache_optimizer_state = None
        else:
            self.register_buffer("hashtbl", torch.empty(0, device=device, dtype=torch.int64))
            self.register_buffer("cache_state", torch.empty(0, device=device, dtype=torch.int32))
            self.cache_optimizer_state = None
            self.cache_weight = None
        self.warmup = True

    # ------------------------------------------------------------------ init
    def full_weight(self) -> torch.Tensor:
        assert self.num_tables == 1, "full_weight() only supported for num_tables == 1 for now"
        return tt_matrix_to_full(self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks, list(self.tt_cores), [1, 0, 2, 3])

    def _assign(self, t: int, values: np.ndarray) -> None:
        core = self.tt_cores[t]
        with torch.no_grad():
            core.copy_(torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32)).reshape(core.shape))

    def reset_parameters(self, weight_dist: str) -> None:
        """The five initialisations of the reference (:613-792)."""
        nd, E, D = self.tt_ndim, self.num_embeddings, self.embedding_dim
        if weight_dist == "uniform":
            sigma = math.sqrt(2.0 / (E + D))
            shrink = float(np.prod(np.asarray(self.tt_ranks, dtype=np.float64) ** (-1.0 / (2 * nd))))
            hi = sigma ** (1.0 / nd) * shrink
            for core in self.tt_cores:
                nn.init.uniform_(core, 0.0, hi)
        elif weight_dist == "naive-uniform":
            for core in self.tt_cores:
                nn.init.uniform_(core, 0.0, 1.0 / math.sqrt(E))
        elif weight_dist == "normal":
            for core in self.tt_cores:
                nn.init.normal_(core, 0.0, 1.0 / math.sqrt(E))
                with torch.no_grad():
                    core.mul_(1.0 / self.tt_ranks[0])
        elif weight_dist == "approx-normal":
            # N(0,1) truncated to |x| >= 2 by rejection, scaled by (3E)^(-1/6).  The reference redraws entry after
            # entry (:650-654): the j-th rejected entry ends up with the j-th accepted value of the stream that
            # follows the first draw.  Done here in blocks -- each block draws exactly as many values as are still
            # missing, so the stream is consumed to the same point and every entry gets the same value.
            scale = (1.0 / math.sqrt(3 * E)) ** (1.0 / 3.0)
            for t, core in enumerate(self.tt_cores):
                w = np.random.normal(0.0, 1.0, size=tuple(core.shape)).astype(np.float32).ravel()
                small = np.flatnonzero(np.abs(w) < 2)
                done = 0
                while done < small.size:
                    x = np.random.normal(0.0, 1.0, size=small.size - done).astype(np.float32)
                    x = x[np.abs(x) >= 2]
                    w[small[done:done + x.size]] = x
                    done += x.size
                self._assign(t, w * np.float64(scale))  # (float64 product, rounded once: `W *= scale` with a numpy scalar)
        elif weight_dist == "approx-uniform":
            self._init_approx_uniform()
        else:
            raise AssertionError(f"unknown weight_dist {weight_dist!r}")

    def _init_approx_uniform(self, sigma: float = 0.01, gridpts: int = 15, width: float = 0.7 / 30.0) -> None:
        """'flat saw-tooth' construction (:660-792): the product of the three
        cores is a train of narrow teeth j/gridpts + U(-width/2, width/2) that
        a narrow Gaussian smears into a uniform density.  3 cores, 1 table."""
        assert self.tt_ndim == 3, "approx-uniform needs exactly 3 TT cores"
        assert self.num_tables == 1, "approx_uniform only supported for num_tables == 1"
        r, p, q = self.tt_ranks, self.tt_p_shapes, self.tt_q_shapes
        scale = 1.0 / (math.sqrt(self.num_embeddings) ** (1.0 / 3.0))

        def teeth(count: int) -> np.ndarray:
            j = np.random.randint(-(gridpts - 1), gridpts, count)
            return j * (1.0 / gridpts) + (-width / 2.0 + width * np.random.rand(count))

        # head: all entries ~ N(1/sqrt(r1), sigma)
        head = (1.0 / math.sqrt(r[1])) + np.random.randn(r[0], p[0], q[0], r[1]) * sigma
        # middle: ~ N(1/sqrt(r1), sigma); per (m, n) one even column k is made tiny except one tooth entry
        mid_scale = 1.0 / math.sqrt(r[1])
        mid = (mid_scale + np.random.randn(r[1], p[1] * q[1], r[2]) * sigma)
        vals = teeth(p[1] * q[1]) / mid_scale
        for ell in range(p[1] * q[1]):
            k = random.randrange(0, r[2], 2)
            mid[:, ell, k] = np.random.randn(r[1]) * (sigma * sigma / mid_scale)
            mid[random.randrange(r[1]), ell, k] = vals[ell]
        mid = mid.reshape(r[1], p[1], q[1], r[2])
        # tail: small background, per (m, n) one odd row carries a tooth
        tail = (np.random.randn(r[2], p[2] * q[2]) * sigma)
        vals = teeth(p[2] * q[2])
        for ell in range(p[2] * q[2]):
            tail[random.randrange(1, r[2], 2) if r[2] > 1 else 0, ell] = vals[ell]
        tail = tail.reshape(r[2], p[2], q[2], r[3])
        for t, w in enumerate((head, mid, tail)):
            self._assign(t, (w * scale).transpose(1, 0, 2, 3).reshape(1, p[t], -1))

    # ----------------------------------------------------------------- cache
    def reset_cache(self) -> None:
        if self.use_cache:
            self.hashtbl.fill_(-1)
            self.cache_freq.fill_(0)
            self.cache_state.fill_(-1)
            self.warmup = True
            self._drop_prefetched(counted=False)  # (the frequency table is empty again)

    @torch.no_grad()
    def _write_back(self, lr: float, steps: int = 1) -> None:
        """`steps` SGD steps of size `lr` on the cores toward every CACHED row (see cache_populate).  Plain SGD whatever the
        module's optimizer (a correction of the cores toward the cached rows, not a training step: Adagrad's accumulators are not
        touched); batches planned ahead hold no plan of the cores' VALUES (plans are index work), so they stay valid."""
        slots = torch.nonzero(self.cache_state >= 0).flatten()
        n = int(slots.numel())
        if n == 0:
            return
        keys = self.hashtbl[slots].contiguous()
        target = self.cache_weight.detach()[self.cache_state[slots].long()]           # what the cached rows have learnt
        offsets = torch.arange(0, n + 1, dtype=torch.int64, device=keys.device)     # one lookup per bag
        no_tbl = torch.empty(0, dtype=torch.int64, device=keys.device)
        colidx, rowidx, tableidx, _, _ = _engine.preprocess_indices_sync(keys, offsets, 1, True, no_tbl,
                                                                          torch.empty(0, dtype=torch.int32, device=keys.device))
        cores = [c.detach() for c in self.tt_cores]
        for _ in range(max(1, int(steps))):
            rows = _engine.tt_forward(1000, 1, n, self.embedding_dim, self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks, self.L, n,
                                      colidx, rowidx, tableidx, cores)                  # the rows as the cores hold them
            diff = (rows - target.unsqueeze(0)).contiguous()  # d/d row of 1/2 ||TT row - cached row||^2
            _engine.tt_sgd_backward(1000, self.embedding_dim, float(lr), self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks, self.L,
                                    n, colidx, rowidx, tableidx, diff, cores)

    def cache_populate(self, write_back: float = 0.0, write_back_steps: int = 1) -> None:
        """tt_embeddings_ops.py:800-814.  `write_back` (NOT in the reference; default 0 = its behaviour): cached rows are trained
        directly, and a populate decompresses every cache row anew from the cores -- what the cached copies learnt since the last
        populate is dropped.  A dense row has no exact TT representation, so there is no inverse; with write_back = lr > 0 the
        cores first take `write_back_steps` fused SGD steps of that size toward the cached rows (gradient of 1/2 ||TT row - cached row||^2
        for every cached key, through the ordinary forward / backward contraction), which moves the TT rows toward what the cache
        learnt -- an approximation (the cached keys share core slices: a few steps recover part of the difference, measured in
        tests/test_module_cpu.py), off by default."""
        if self.use_cache and write_back > 0.0 and not self.warmup:
            self._write_back(write_back, write_back_steps)
        if self.use_cache:
            extra = {"reference_exact": True} if self.reference_exact_populate else {}
            _engine.cache_populate(self.num_embeddings, self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks,
                                   list(self.tt_cores), self.L, self.hashtbl, self.cache_freq, self.cache_state,
                                   self.cache_weight, **extra)
            self.warmup = False
            self._drop_prefetched()  # (a planned-ahead batch was split into hits / misses by the OLD cache contents)

    def update_cache(self, indices: torch.Tensor) -> None:
        if self.use_cache:
            _engine.update_cache_state(indices, self.hashtbl, self.cache_freq)

    # -------------------------------------------------------------- prefetch
    def _normalise(self, indices: torch.Tensor, offsets: torch.Tensor):
        indices, offsets = indices.long(), offsets.long()
        if not self.include_last_offset:
            offsets = torch.cat([offsets, offsets.new_full((1,), indices.numel())])
        return indices, offsets

    def prefetch(self, indices: torch.Tensor, offsets: torch.Tensor) -> bool:
        """Not in the reference.  Run the lookup prologue of a COMING batch now, on a side stream: hash-table frequency
        update, offsets -> bag rows and the lookup plan depend on the batch's indices only, not on the cores, so they
        can overlap the backward of the step before (call it once the next batch's tensors exist, before
        `loss.backward()`; the next `forward(indices, offsets)` with the same tensor OBJECTS, unmodified, picks the
        result up -- anything else runs the prologue in line).  HIP streams
        and events only; captures into a hipGraph as a forked branch.  (Host cost of a call: ~25 us of stream / event
        handling -- worth it inside a captured step or for steps beyond ~0.1 ms; an eager loop of small steps is better served
        by prefetch_many(), one launch for a round of batches.)  Returns False (and does nothing) whenever the
        overlap does not apply: cache live, no C++ node, CPU tensors, empty batch, duplicate sharing."""
        fast = _native_node()
        if (fast is None or not self.warmup or not indices.is_cuda or indices.numel() == 0 or self._dedup_may_share(indices.numel())
                or indices.dim() != 1 or offsets.dim() != 1 or self.__dict__.get("_split0", 0) > 1):
            return False
        key = (id(indices), id(offsets))  # (identity: the entry keeps both objects alive, so neither id nor memory is reused)
        idx, off = self._normalise(indices, offsets)
        idx, off = idx.contiguous(), off.contiguous()
        cur = torch.cuda.current_stream(indices.device)
        side = self.prefetch_stream(indices.device)
        ready = torch.cuda.Event()
        ready.record(cur)  # the batch's tensors exist at this point of the caller's stream
        if side != cur:  # made (perhaps cast / concatenated) on the caller's stream, read on the side stream
            idx.record_stream(side)
            off.record_stream(side)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            rowidx, tableidx, plan = fast.prologue(idx, off, self.num_tables, getattr(self, "_p_flat", self.tt_p_shapes),
                                                   self.tt_q_shapes, self.tt_ranks,
                                                   self.hashtbl if self.use_cache else None,
                                                   self.cache_freq if self.use_cache else None)
            done = torch.cuda.Event()
            done.record(side)
        if len(self._prefetched) >= 8:  # (batches that never came: drop the oldest)
            self._evict_oldest_prefetched()
        self._prefetched[key] = (idx, off, (rowidx, tableidx, plan), done, indices, offsets, indices._version,
                                 offsets._version, False)
        return True

    def prefetch_many(self, batches) -> bool:
        """Not in the reference.  The lookup prologues of SEVERAL coming batches -- `[(indices, offsets), ...]`, all of
        one size -- in one launch on the current stream (`ttx_lookup_prologue_multi`): a prologue occupies 30 of the chip's
        256 CUs for its ~12 us of dependent loads, sixteen of them take about as long as one.  Each batch's next
        `forward(indices, offsets)` picks its result up, as after prefetch().  With a LIVE cache (one table) the round's
        frequency updates, cache lookups, hit / miss partitions and miss plans are done the same way
        (`ttx_lookup_prologue_cached_multi`, three launches); cache_populate() / reset_cache() drop what was planned.
        Returns False (and does nothing) where the prologue cannot be planned ahead: no C++ node, CPU tensors, empty
        batches, batches of different sizes, duplicate sharing, a live cache over several tables."""
        fast = _native_node()
        batches = list(batches)
        live = not self.warmup
        if fast is None or not batches or self.__dict__.get("_split0", 0) > 1 or self._dedup_may_share(batches[0][0].numel()) or (live and not (self.use_cache and self.num_tables == 1)):
            return False
        norm, keys = [], []
        for indices, offsets in batches:
            if not indices.is_cuda or indices.numel() == 0 or indices.dim() != 1 or offsets.dim() != 1:
                return False
            keys.append((id(indices), id(offsets)))
            idx, off = self._normalise(indices, offsets)
            norm.append((idx.contiguous(), off.contiguous()))
        if len({(i.numel(), o.numel()) for i, o in norm}) != 1:
            return False
        self.prefetch_stream(norm[0][0].device)
        if live:
            res = fast.prologue_cached_multi([i for i, _ in norm], [o for _, o in norm], self.tt_p_shapes, self.tt_q_shapes,
                                             self.tt_ranks, self.hashtbl, self.cache_freq, self.cache_state)
        else:
            res = fast.prologue_multi([i for i, _ in norm], [o for _, o in norm], self.num_tables,
                                      getattr(self, "_p_flat", self.tt_p_shapes), self.tt_q_shapes, self.tt_ranks,
                                      self.hashtbl if self.use_cache else None,
                                      self.cache_freq if self.use_cache else None)
        for k, key in enumerate(keys):
            while len(self._prefetched) >= 64:
                self._evict_oldest_prefetched()
            self._prefetched[key] = (norm[k][0], norm[k][1], tuple(t[k] for t in res), None, batches[k][0], batches[k][1],
                                     batches[k][0]._version, batches[k][1]._version, live)
        return True

    def _cached_args_stale(self, cores, state, buffers) -> bool:
        """the cached argument lists (_fa / _fc) hold the Parameter / buffer OBJECTS; anything that re-binds one --
        load_state_dict(assign=True), m.tt_cores[i] = nn.Parameter(..), register_buffer over an old name -- must not leave the
        native node training the orphaned tensors.  Identity checks against the module's own dicts: no attribute protocol, ~1 us."""
        mods = self._modules  # (the sub-modules' own dicts, reached without nn.Module.__getattr__)
        pc, ps = mods["tt_cores"]._parameters, mods["optimizer_state"]._buffers  # (BufferList: registered buffers, in order)
        if len(pc) != len(cores) or not all(map(_is, cores, pc.values())):
            return True
        if len(ps) != len(state) or not all(map(_is, state, ps.values())):
            return True
        bufs = self._buffers
        return any(t is not None and bufs.get(name) is not t for name, t in buffers)

    def _apply(self, fn, *a, **kw):
        """module.to() / .cuda() / .float(): buffers are replaced by new tensors -- forget the cached argument lists"""
        self.__dict__.pop("_fa", None)
        self.__dict__.pop("_fc", None)
        self.__dict__.pop("_sh0", None)  # (the padded copy of core 0 belongs to the old tensors' device / dtype)
        return super()._apply(fn, *a, **kw)

    def _evict_oldest_prefetched(self) -> None:
        """drop the oldest planned-ahead batch; its frequency update has been issued, so should the batch come after
        all, its in-line prologue must not count it again (forward looks it up in _pf_evicted)"""
        k = next(iter(self._prefetched))
        hit = self._prefetched.pop(k)
        if self.use_cache:
            ev = self.__dict__.setdefault("_pf_evicted", {})
            while len(ev) >= 64:
                ev.pop(next(iter(ev)))
            ev[k] = (hit[4], hit[5], hit[6], hit[7])

    def _drop_prefetched(self, counted: bool = True) -> None:
        """forget what was planned ahead.  counted: the dropped batches' frequency updates stay in the table
        (cache_populate), so their in-line prologues must not repeat them."""
        while counted and getattr(self, "_prefetched", None):
            self._evict_oldest_prefetched()
        if getattr(self, "_prefetched", None):
            self._prefetched.clear()
        if not counted and getattr(self, "_pf_evicted", None):
            self._pf_evicted.clear()

    def __getstate__(self):
        """copy.deepcopy / torch.save of the module: the prefetch side stream and the planned-ahead batches (device
        buffers, HIP events) belong to this process and this point in time -- they are recreated at first use."""
        state = self.__dict__.copy()
        for k in ("_pf_stream", "_prefetched", "_pf_key", "_pf_counted", "_pf_evicted", "_fa", "_fc"):
            state.pop(k, None)
        return state

    def prefetch_stream(self, device: Optional[torch.device] = None) -> "torch.cuda.Stream":
        """the side stream of prefetch() (created at first use; call this before a hipGraph capture that prefetches)"""
        side = getattr(self, "_pf_stream", None)
        if side is None:
            side = self._pf_stream = torch.cuda.Stream(device=device if device is not None else self.tt_cores[0].device)
            self._prefetched = {}
        return side

    def _take_prefetched(self, indices: torch.Tensor, offsets: torch.Tensor, live: bool = False):
        """the planned-ahead prologue of this batch -- (rowidx, tableidx, plan), or with a live cache (tableidx, pcol, prow,
        ploc, n_tt, plan) -- or None"""
        d = self.__dict__  # (plain attributes, read and written past nn.Module's attribute protocol: ~1 us apiece on the host-bound step)
        pf = d.get("_prefetched")
        if not pf:
            return None
        hit = pf.pop(d.get("_pf_key"), None)
        d["_pf_key"] = None
        if hit is None:
            return None
        if hit[8] != live:  # planned for the other state of the cache: the prologue runs in line -- without counting
            d["_pf_counted"] = self.use_cache  # the batch into the frequency table a second time
            return None
        pre, done = hit[2], hit[3]
        if done is not None:  # (prefetch(): ran on the side stream; prefetch_many(): same stream, stream-ordered)
            cur = torch.cuda.current_stream(indices.device)
            cur.wait_event(done)
            for t in pre:  # allocated on the side stream, used (and freed) on this one
                t.record_stream(cur)
        return pre

    # --------------------------------------------------------------- forward
    def _forward_split0(self, fast, indices: torch.Tensor, offsets: torch.Tensor, k: int) -> torch.Tensor:
        """q0 = k q0': k part lookups per index through the C++ node on the table [k p0, p1, p2] x [q0', q1, q2] (views of
        core 0 and its optimizer state: the same memory); the output [tables, k B, D / k] IS [tables, B, D]."""
        use_state = self.sparse and self.optimizer not in _SGD_LIKE
        optim = 2 if not self.sparse else (1 if use_state else 0)
        if self.use_cache:  # (the frequency table counts the caller's indices, not the part lookups)
            _engine.update_cache_state(indices, self.hashtbl, self.cache_freq)
        p, q, nt = self.tt_p_shapes, self.tt_q_shapes, self.num_tables
        vi, vo = _engine.split0_expand(indices, offsets, k, p[1] * p[2])
        Q = self.__dict__.get("_pad0", 0) or q[0]
        c0 = self.tt_cores[0]
        s0 = self.optimizer_state[0] if use_state else None
        if Q != q[0]:  # core 0 (and its optimizer state) zero-padded to Q slots: `_pad0_target`
            c0, s0 = self._padded0(Q, optim, use_state)
        cores = [c0.view(nt, k * p[0], c0.size(2) // k), self.tt_cores[1], self.tt_cores[2]]
        state = [s0.view(nt, k * p[0], s0.size(2) // k), self.optimizer_state[1], self.optimizer_state[2]] if use_state else []
        out = fast.lookup(vi, vo, nt, [k * p[0], p[1], p[2]], [Q // k, q[1], q[2]], self.tt_ranks, optim,
                          self.learning_rate, self.eps, None, None, state, cores, None, None, None, None)
        B = (offsets.numel() - 1) // nt
        if Q == q[0]:
            return out.view(nt, B, self.embedding_dim)
        if optim != 2 and out.grad_fn is not None:
            # the fused optimizer has updated the padded copies when this node's backward returns: the real slots go back
            # into the module's own core 0 / state 0 (what state_dict(), full_weight() and every other route read)
            out.grad_fn.register_hook(lambda gi, go: self._padded0_store())
        # (a copy, not the strided prefix view: callers of the reference's module get a contiguous [tables, B, D])
        return out.view(nt, B, Q * q[1] * q[2])[:, :, :self.embedding_dim].contiguous()

    def _padded0(self, Q: int, optim: int, use_state: bool):
        """core 0 [tables, p0, q0 r1] zero-padded to [tables, p0, Q r1] (and the same of its optimizer state).  Dense gradients
        (sparse=False): an autograd pad of the Parameter, every step.  Fused optimizers: a buffer the module keeps, refreshed from
        the Parameter (and the optimizer state) EVERY step -- p0 q0 r1 floats, one small copy.  (Round 4 refreshed only when the
        Parameter's `_version` moved or it was re-bound; writes through `.data` -- p.data.mul_(..), EMA / averaging code -- do not
        move the version, the next step then trained the stale copy and the write-back hook overwrote the caller's update: round 4
        advisor.)"""
        c, w = self.tt_cores[0], (Q - self.tt_q_shapes[0]) * self.tt_ranks[1]
        if optim == 2:
            return torch.nn.functional.pad(c, (0, w)), None
        sh = self.__dict__.get("_sh0")
        if sh is None or sh[0].device != c.device:
            mk = lambda: torch.zeros(c.size(0), c.size(1), c.size(2) + w, device=c.device, dtype=c.dtype)  # noqa: E731
            sh = self._sh0 = [mk().requires_grad_(True), mk() if use_state else None, None, -1, None, -1]
        srcs = [(0, c, 2)] + ([(1, self.optimizer_state[0], 4)] if use_state else [])
        with torch.no_grad():
            for j, src, at in srcs:
                if sh[j] is None:
                    sh[j] = torch.zeros_like(sh[0])
                sh[j][:, :, :src.size(2)].copy_(src)  # (the padded slots stay zero: nothing ever writes a non-zero there)
                sh[at], sh[at + 1] = src, src._version
        return sh[0], sh[1]

    @torch.no_grad()
    def _padded0_store(self) -> None:
        sh = self._sh0
        for j, at in ((0, 2), (1, 4)):
            src = sh[at]
            if src is not None and sh[j] is not None:
                src.copy_(sh[j][:, :, :src.size(2)])
                sh[at + 1] = src._version

    def _forward_n(self, indices: torch.Tensor, offsets: torch.Tensor, n_dev: torch.Tensor) -> torch.Tensor:
        """forward with a device-side lookup count (see forward): the prologue is built for the live lookups (C++ node:
        prologue(n_dev=) -> lookup(pre_*=)); routes that cannot take a device count read it back and slice."""
        fast = _native_node()
        plain = (fast is not None and self.warmup and indices.is_cuda and indices.numel() > 0 and not self.use_cache
                 and self.__dict__.get("_split0", 0) <= 1 and not self._dedup_may_share(indices.numel()))
        if not plain:  # (CPU tensors under the tests' oracle engine, a cache, part lookups, shared duplicates)
            if indices.is_cuda and torch.cuda.is_current_stream_capturing():
                # (round 6, advisor: this route reads the count back -- say so instead of failing inside the capture)
                raise RuntimeError("TableBatchedTTEmbeddingBag.forward(n_dev=): with a cache, part lookups (q0 > 4) or shared duplicates "
                                   "the live count is read back to the host -- not capturable in a hipGraph on this route")
            n = int(n_dev.item())
            return self.forward(indices[:n].contiguous(), offsets)
        indices = indices.long() if indices.dtype != torch.int64 else indices
        offsets = offsets.long() if offsets.dtype != torch.int64 else offsets
        indices = indices if indices.is_contiguous() else indices.contiguous()
        offsets = offsets if offsets.is_contiguous() else offsets.contiguous()
        use_state = self.sparse and self.optimizer not in _SGD_LIKE
        optim = 2 if not self.sparse else (1 if use_state else 0)
        p_flat = getattr(self, "_p_flat", self.tt_p_shapes)
        pre = fast.prologue(indices, offsets, self.num_tables, p_flat, self.tt_q_shapes, self.tt_ranks, None, None,
                            n_dev.to(torch.int32).reshape(1))
        return fast.lookup(indices, offsets, self.num_tables, p_flat, self.tt_q_shapes, self.tt_ranks, optim,
                           self.learning_rate, self.eps, None, None, list(self.optimizer_state) if use_state else [],
                           list(self.tt_cores), None, *pre)

    def _det_bits(self) -> int:
        """the C++ node's encoding of deterministic_cache_update (csrc/ttx_torch.cpp: bit 9 sorted, bit 10 atomics, neither auto)"""
        det = self.__dict__.get("deterministic_cache_update")
        if det is None and os.environ.get("TTX_DETERMINISTIC", "") != "":
            det = os.environ["TTX_DETERMINISTIC"] not in ("0", "")
        return 0 if det is None else (512 if det else 1024)

    def _dedup_may_share(self, nnz: int) -> bool:
        d = getattr(self, "dedup", False)
        return bool(d) and (d != "auto" or nnz >= _DEDUP_AUTO_MIN_NNZ)

    def _dedup_auto(self, nnz: int):
        """dedup="auto": (share this batch?, sample its distinct fraction?)."""
        if nnz < _DEDUP_AUTO_MIN_NNZ:
            return False, False
        st = self.__dict__.setdefault("_dd_auto", [False, 0])
        if st[1] <= 0 and not torch.cuda.is_current_stream_capturing():
            return True, True
        st[1] -= 1
        return st[0], False

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, warmup: bool = True,
                per_sample_weights: Optional[torch.Tensor] = None, n_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> [num_tables, B, D].  (`warmup` is ignored like in the reference,
        which uses self.warmup, :822,:841.)  int32 indices / offsets are accepted; with
        include_last_offset=False the closing offset (nnz) is appended here.
        `n_dev` (round 5; not in the reference): one int32 on the device, the number of LIVE lookups -- `indices` then holds an
        upper bound (a fixed-capacity buffer), `offsets` (with its closing entry) describes exactly the first n_dev of them, and
        only those are planned, contracted, pooled and trained: no host read-back of the count (ttx_lookup_prologue_n).  What
        the table-sharded module's ragged route hands its local lookup."""
        if n_dev is not None:
            if per_sample_weights is not None:  # (round 6, advisor: the weights were silently dropped on this route)
                raise NotImplementedError("forward(n_dev=) does not take per_sample_weights")
            if not self.include_last_offset:
                raise ValueError("forward(n_dev=) needs offsets with their closing entry (include_last_offset=True): the closing "
                                 "offset IS the live count, which only the device knows")
            return self._forward_n(indices, offsets, n_dev)
        if per_sample_weights is not None:
            # nn.EmbeddingBag(mode="sum") semantics: forward, the cores' gradients / fused optimizers and -- if the
            # weights require it -- their own gradient.  Served by the C++ nodes: cache not live, or live over one table
            # (the weights follow their lookups through the hit / miss partition; cache rows are gathered and updated
            # weighted).
            if _native_node() is None or not indices.is_cuda or (not self.warmup and not (self.use_cache and self.num_tables == 1)):
                raise NotImplementedError("per_sample_weights needs ttx_torch.so (GPU tensors)")
            if per_sample_weights.shape != indices.shape:
                raise ValueError("per_sample_weights must have the shape of indices")
            per_sample_weights = per_sample_weights.float().contiguous()  # (keeps the autograd graph of the weights)
        if indices.dim() != 1 or offsets.dim() != 1:
            raise ValueError("indices and offsets must be 1-D (the 2-D fixed-length form of nn.EmbeddingBag is not supported)")
        d = self.__dict__  # (plain attributes past nn.Module's __setattr__ / __getattr__: 1.3 / 0.8 us apiece, five per call)
        d["_pf_key"] = None
        d["_pf_counted"] = False  # this batch's frequency update was issued by a planned-ahead prologue that is not used
        if d.get("_prefetched"):  # a prefetch() for exactly these tensor objects, not written to since?
            k = (id(indices), id(offsets))
            hit = self._prefetched.get(k)
            if hit is not None and (hit[4] is not indices or hit[5] is not offsets or hit[6] != indices._version
                                    or hit[7] != offsets._version):
                self._prefetched.pop(k)  # stale: the batch was modified in place after its prefetch
                hit = None
            if hit is not None:
                d["_pf_key"] = k
                indices, offsets = hit[0], hit[1]  # (already in the int64 / closing-offset form)
        if self._pf_key is None:
            ev = d.get("_pf_evicted")
            if ev:
                e = ev.pop((id(indices), id(offsets)), None)
                if e is not None and e[0] is indices and e[1] is offsets and e[2] == indices._version and e[3] == offsets._version:
                    d["_pf_counted"] = True
            indices, offsets = self._normalise(indices, offsets)
        if (offsets.numel() - 1) % self.num_tables != 0:
            raise ValueError(f"offsets must describe num_tables * B bags, got {offsets.numel() - 1} bags for "
                             f"{self.num_tables} tables")
        fast = _native_node()
        share = getattr(self, "dedup", False) and self.warmup and indices.is_cuda and indices.numel() > 0 \
            and per_sample_weights is None and getattr(_engine, "DedupPlan", None) is not None
        sample = False
        if share and self.dedup == "auto":
            share, sample = self._dedup_auto(indices.numel())
        if share:
            # duplicate lookups share their contraction: frequency update + bag rows as usual, then the map and the
            # plan of the distinct pairs (one work-group sorts the batch's keys), through the reference-shaped route
            indices, rowidx, tableidx, n_tt, cache_locations = _engine.preprocess_indices_sync(
                indices, offsets, self.num_tables, True, self.hashtbl, self.cache_state,
                *((self.cache_freq,) if self.use_cache else ()))
            rowidx._ttx_plan = _engine.make_plan(self.num_tables, self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks,
                                                 n_tt, indices, tableidx, rowidx, dedup=True)
            if sample:  # the map's header holds the number of distinct pairs (one read-back per sampling period)
                dd = getattr(rowidx._ttx_plan, "dd", None)
                frac = float(dd[:4].view(torch.int32).item()) / max(n_tt, 1) if dd is not None else 1.0
                self._dd_auto = [frac <= _DEDUP_AUTO_MAX_DISTINCT, _DEDUP_AUTO_PERIOD]
                self._dd_auto_last = frac
            return TTLookupFunction.apply(
                (offsets.numel() - 1) // self.num_tables, self.embedding_dim, self.tt_p_shapes, self.tt_q_shapes,
                self.tt_ranks, self.L, n_tt, 0, indices, rowidx, tableidx, self.optimizer, self.learning_rate,
                self.eps, self.sparse, None, self.cache_optimizer_state, self.cache_weight,
                list(self.optimizer_state), *self.tt_cores)
        if (self.__dict__.get("_split0", 0) > 1 and fast is not None and self.warmup and indices.is_cuda and indices.numel() > 0
                and per_sample_weights is None and not hasattr(self, "_p_flat")):
            return self._forward_split0(fast, indices, offsets, self._split0)
        if fast is not None and self.warmup and indices.is_cuda and indices.numel() > 0:
            # cache not live: the whole lookup (prologue, forward, and the backward / fused optimizer node)
            # is the C++ autograd node of csrc/ttx_torch.cpp -- same C ABI calls, no interpreter in between
            use_state = self.sparse and self.optimizer not in _SGD_LIKE
            optim = 2 if not self.sparse else (1 if use_state else 0)
            pre = self._take_prefetched(indices, offsets)  # (rowidx, tableidx, plan) if prefetch() ran for this batch
            count = self.use_cache and not self._pf_counted
            fa = self.__dict__.get("_fa")  # (cores, state, hashtbl, cache_freq, p): looked up once, not per step -- every
            if fa is not None and self._cached_args_stale(fa[0], fa[1], (("hashtbl", fa[2]), ("cache_freq", fa[3]))):
                fa = None                  #  (a Parameter / buffer was re-bound: load_state_dict(assign=True), tt_cores[i] = ..)
            if fa is None:                 #  nn.Module attribute / ParameterList access is microseconds of a host-bound step
                fa = self._fa = (list(self.tt_cores), list(self.optimizer_state), self.hashtbl if self.use_cache else None,
                                 self.cache_freq if self.use_cache else None,
                                 getattr(self, "_p_flat", self.tt_p_shapes))  # (per-table factors: flattened)
            out = fast.lookup(indices if indices.is_contiguous() else indices.contiguous(),
                              offsets if offsets.is_contiguous() else offsets.contiguous(), self.num_tables, fa[4],
                              self.tt_q_shapes, self.tt_ranks, optim, self.learning_rate, self.eps,
                              fa[2] if count else None, fa[3] if count else None,
                              fa[1] if use_state else [], fa[0], per_sample_weights,
                              *(pre if pre is not None else (None, None, None)))
            if optim != 2 and _DIRECT_BACKWARD and out.requires_grad and (per_sample_weights is None or not per_sample_weights.requires_grad):
                _direct_register(out, fast.node_of(out))  # (fused optimizer: backward() of this tensor may skip autograd's engine)
            return out
        if (fast is not None and not self.warmup and self.use_cache and self.num_tables == 1 and indices.is_cuda
                and indices.numel() > 0):
            # cache live: frequency update + hash lookup + stable partition (split point kept on the device),
            # contraction of the misses, gather of the hits, and the matching backward -- the C++ node again
            use_state = self.sparse and self.optimizer not in _SGD_LIKE
            optim = 2 if not self.sparse else (1 if use_state else 0)
            pre = self._take_prefetched(indices, offsets, live=True)  # planned ahead by prefetch_many()?
            if per_sample_weights is not None and pre is not None:
                pre = None  # (the planned-ahead partition does not carry weights: this batch's prologue runs in line,
                self._pf_counted = True  # without counting the batch a second time)
            fc = self.__dict__.get("_fc")  # (the module's tensors, looked up once: see _fa above)
            if fc is not None and self._cached_args_stale(fc[0], fc[1], (("hashtbl", fc[2]), ("cache_freq", fc[3]), ("cache_state", fc[4]),
                                                                         ("cache_optimizer_state", fc[5]), ("cache_weight", fc[6]))):
                fc = None
            if fc is None:
                fc = self._fc = (list(self.tt_cores), list(self.optimizer_state), self.hashtbl, self.cache_freq, self.cache_state,
                                 self.cache_optimizer_state, self.cache_weight)
            out = fast.lookup_cached(indices if indices.is_contiguous() else indices.contiguous(),
                                     offsets if offsets.is_contiguous() else offsets.contiguous(), self.tt_p_shapes, self.tt_q_shapes,
                                     self.tt_ranks, optim | (256 if self._pf_counted else 0) | self._det_bits(), self.learning_rate, self.eps,
                                     fc[2], fc[3], fc[4], fc[5] if use_state else None,
                                     fc[6], fc[1] if use_state else [],
                                     fc[0], list(pre) if pre is not None else [], per_sample_weights)
            if optim != 2 and _DIRECT_BACKWARD and out.requires_grad and (per_sample_weights is None or not per_sample_weights.requires_grad):
                _direct_register(out, fast.node_of(out))
            return out
        prologue = getattr(_engine, "lookup_prologue", None)
        if prologue is not None and self.warmup and indices.numel() > 0:
            # cache not live: frequency update, offsets -> bag rows and the lookup plan in one native call
            rowidx, tableidx, plan = prologue(indices, offsets, self.num_tables, self.tt_p_shapes, self.tt_q_shapes,
                                              self.tt_ranks, self.hashtbl if self.use_cache else None,
                                              self.cache_freq if self.use_cache else None)
            rowidx._ttx_plan = plan  # picked up by TTLookupFunction.forward
            n_tt, cache_locations = indices.numel(), None
        elif self.use_cache and getattr(_engine, "FUSED_CACHE_UPDATE", False) and indices.numel() > 0:
            # frequency update folded into the preprocessing launch (same order as the reference:
            # count the batch's indices, then look them up)
            indices, rowidx, tableidx, n_tt, cache_locations = _engine.preprocess_indices_sync(
                indices, offsets, self.num_tables, self.warmup, self.hashtbl, self.cache_state, self.cache_freq)
        else:
            self.update_cache(indices)
            indices, rowidx, tableidx, n_tt, cache_locations = _engine.preprocess_indices_sync(
                indices, offsets, self.num_tables, self.warmup, self.hashtbl, self.cache_state)
        n_cached = indices.numel() - n_tt
        det = self.__dict__.get("deterministic_cache_update")
        if det is not None and n_cached > 0:
            rowidx._ttx_det = bool(det)  # (picked up by TTLookupFunction.forward, like the plan: the reference's 19 arguments stay)
        return TTLookupFunction.apply(
            (offsets.numel() - 1) // self.num_tables, self.embedding_dim, self.tt_p_shapes, self.tt_q_shapes,
            self.tt_ranks, self.L, n_tt, n_cached, indices, rowidx, tableidx, self.optimizer, self.learning_rate,
            self.eps, self.sparse, cache_locations, self.cache_optimizer_state, self.cache_weight,
            list(self.optimizer_state), *self.tt_cores)

    def set_learning_rate(self, lr: float) -> None:
        self.learning_rate = lr

    def get_params(self) -> List[torch.Tensor]:
        params = list(self.tt_cores)
        if self.use_cache:
            params.append(self.cache_weight)
        return params


class TTEmbeddingBag(TableBatchedTTEmbeddingBag):
    """TT embedding bag for exactly one table; forward returns [B, D]."""

    def __init__(self, num_embeddings: int, embedding_dim: int, tt_ranks: List[int],
                 tt_p_shapes: Optional[List[int]] = None, tt_q_shapes: Optional[List[int]] = None,
                 optimizer: OptimType = OptimType.SGD, learning_rate: float = 0.1, eps: float = 1.0e-10,
                 sparse: bool = True, use_cache: bool = True, cache_size: int = 0, hashtbl_size: int = 0,
                 weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False,
                 device: Optional[torch.device] = None, include_last_offset: bool = True, dedup: bool = False,
                 reference_exact_populate: bool = False, deterministic_cache_update: Optional[bool] = None) -> None:
        super().__init__(1, num_embeddings, embedding_dim, tt_ranks, tt_p_shapes, tt_q_shapes, optimizer,
                         learning_rate, eps, sparse, use_cache, cache_size, hashtbl_size, weight_dist,
                         enforce_embedding_dim, device, include_last_offset, dedup, reference_exact_populate,
                         deterministic_cache_update)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, warmup: bool = True,
                per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        # squeeze is a view both ways: `[0]` would make autograd materialise a zero [1,B,D] buffer
        # and copy the gradient into it (two extra kernels per step)
        out = super().forward(indices, offsets, warmup, per_sample_weights)
        res = out.squeeze(0)
        e = _direct.get(id(out)) if _direct else None
        if e is not None and e[0]() is out:  # (the lookup's own node: backward() of the view is backward() of the node)
            _direct_register(res, e[1])
        return res
