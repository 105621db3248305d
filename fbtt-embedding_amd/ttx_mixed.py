"""Tables of DIFFERENT cardinality behind one module.

The reference batches only tables of identical shape (`TableBatchedTTEmbeddingBag`
asserts one `num_embeddings` / one set of TT shapes, tt_embeddings_ops.py:424,
README.md:136-141), so a real DLRM -- 26 sparse features with 26 different
cardinalities -- needs 26 separate modules and 26 x the launches.  Here the tables
are grouped by their TT shape: every group is ONE `TableBatchedTTEmbeddingBag`
(one plan, one forward, one backward for all its tables), and the call form is
DLRM's: one (indices, offsets) pair per table in, one [B, D] per table out.
`fused=True` goes one step further: ONE batched lookup for all tables whatever
their row factors (`VarTableTTEmbeddingBag`, include/ttx.h `ttx_geom::p_tables`).
Measured (scripts/bench_mixed.py, 8 tables of 4 shapes, B=512 x 20 lookups, eager
fwd+bwd+SGD): one module per table 0.600 ms/step, grouped 0.506, fused 0.293.
`streams=True` puts every group on a HIP stream of its own; the eager step is
host-bound at these sizes and the stream switches cost what the overlap gives
(0.600 ms/step) -- it is there for large batches and captured graphs.

    emb = MixedTTEmbeddingBag([1460, 583, 10131227, ...], 64, tt_ranks=[32, 32])
    outs = emb(lS_i, lS_o)        # lists of per-table tensors -> list of [B, D]
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from tt_embeddings_ops import OptimType, TableBatchedTTEmbeddingBag, suggested_tt_shapes


def merge_bags(indices: Sequence[torch.Tensor], offsets: Sequence[torch.Tensor],
               include_last_offset: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """per-table (indices, offsets) -> the table-major batched form of `TableBatchedTTEmbeddingBag`
    (offsets with num_tables * B + 1 entries).  Lengths come from tensor shapes: no device read-back."""
    assert len(indices) == len(offsets) and len(indices) > 0
    nb = {int(o.numel()) for o in offsets}
    if len(nb) != 1:
        raise ValueError(f"every table must describe the same number of bags, got offsets of {sorted(nb)} entries")
    if include_last_offset and nb.pop() < 1:
        raise ValueError("include_last_offset form: offsets needs at least the closing entry")
    if __debug__ and include_last_offset and not indices[0].is_cuda:  # (host tensors only: no device read-back)
        for idx, off in zip(indices, offsets):
            assert int(off[-1]) == idx.numel(), "closing offset must equal the number of indices of the table"
    parts, base = [], 0
    for idx, off in zip(indices, offsets):
        starts = off[:-1] if include_last_offset else off
        parts.append(starts + base if base else starts)
        base += int(idx.numel())
    parts.append(torch.full((1,), base, dtype=parts[0].dtype, device=parts[0].device))
    return torch.cat([i.reshape(-1) for i in indices]), torch.cat(parts)


class VarTableTTEmbeddingBag(TableBatchedTTEmbeddingBag):
    """`TableBatchedTTEmbeddingBag` for tables of DIFFERENT row factors (same q and ranks): one plan, one
    forward, one backward launch set for all of them (include/ttx.h `ttx_geom::p_tables`).  Core t is one
    Parameter [1, sum_k p_k_t, r_t q_t r_{t+1}] holding the tables' slices one table after the other
    (`table_rows(t)[k]` = table k's rows of it); forward takes the table-major batched form, like the parent,
    and returns [num_tables, B, D].  No cache (the parent allows one for a single table only)."""

    def __init__(self, num_embeddings: Sequence[int], embedding_dim: int, tt_ranks: List[int],
                 tt_p_shapes: Optional[Sequence[Optional[List[int]]]] = None, tt_q_shapes: Optional[List[int]] = None,
                 optimizer: OptimType = OptimType.SGD, learning_rate: float = 0.1, eps: float = 1.0e-10,
                 sparse: bool = True, weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False,
                 device: Optional[torch.device] = None, include_last_offset: bool = True,
                 table_ranks: Optional[Sequence[List[int]]] = None, table_q: Optional[Sequence[List[int]]] = None) -> None:
        """`table_q` (one factoring per table, entry by entry <= `tt_q_shapes`, every one a factoring of `embedding_dim`; round 4):
        tables whose output rows are factored DIFFERENTLY ride in the same batched lookup.  `tt_q_shapes` is then the common
        (padded) factoring the kernels run -- prod(tt_q_shapes) >= embedding_dim values per padded output row -- table k's cores are
        stored zero-padded to it like the ranks below, and forward() gathers every table's own embedding_dim values out of its
        padded rows (one gather launch; its backward scatters the gradient into zeros, so the padding's gradient is zero and the
        padding stays zero).
        `table_ranks` (one list per table, every entry <= the matching `tt_ranks` entry): tables of SMALLER TT ranks ride in the
        same batched lookup -- table k is initialised as a table of its own ranks and its slices are stored zero-padded to the
        common ranks.  The padding stays zero under the fused optimizers (every gradient term of a padded entry has a zero factor),
        so table k keeps behaving as a rank-`table_ranks[k]` table; what it costs is the multiply-adds on the zeros."""
        nn.Module.__init__(self)
        self.include_last_offset = bool(include_last_offset)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("VarTableTTEmbeddingBag needs a GPU")
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        nd = len(tt_ranks) + 1
        Es = [int(e) for e in num_embeddings]
        assert len(Es) >= 1 and all(e > 0 for e in Es)
        ps = []
        for k, e in enumerate(Es):
            given = tt_p_shapes[k] if tt_p_shapes is not None else None
            ps.append([int(x) for x in given] if given is not None else suggested_tt_shapes(e, nd))
            assert len(ps[-1]) == nd and int(np.prod(np.asarray(ps[-1], dtype=np.int64))) >= e
        self.tt_q_shapes = [int(x) for x in tt_q_shapes] if tt_q_shapes is not None \
            else suggested_tt_shapes(int(embedding_dim), nd, allow_round_up=not enforce_embedding_dim)
        self.table_q = None if table_q is None else [[int(x) for x in q] for q in table_q]
        self.out_dim = int(embedding_dim)  # what forward() returns per bag
        if self.table_q is None:
            assert int(np.prod(self.tt_q_shapes)) == int(embedding_dim)
        else:
            assert tt_q_shapes is not None and len(self.table_q) == len(Es), "table_q: with the common tt_q_shapes, one per table"
            assert all(len(q) == nd and int(np.prod(q)) == self.out_dim and all(a <= b for a, b in zip(q, self.tt_q_shapes))
                       for q in self.table_q), "table_q: per table a factoring of embedding_dim, entry by entry <= tt_q_shapes"
        self.num_tables, self.tt_ndim = len(Es), nd
        # (the parent's forward sizes its output rows by embedding_dim: the PADDED row length when the factorings differ)
        self.table_num_embeddings, self.embedding_dim = Es, int(np.prod(self.tt_q_shapes))
        self.num_embeddings = max(Es)
        self.tt_ranks = [1] + [int(x) for x in tt_ranks] + [1]
        self.tt_p_shapes = ps                                  # one list per table (the ctypes route)
        self._p_flat = [v for row in ps for v in row]          # ... flattened for the C++ node
        self.sparse, self.optimizer, self.learning_rate, self.eps = sparse, optimizer, learning_rate, eps
        self.register_buffer("L", torch.zeros(nd, dtype=torch.int64, device=device))  # (strides are per table)
        from tt_embeddings_ops import _SGD_LIKE, BufferList
        self.tt_cores = nn.ParameterList()
        self.optimizer_state = BufferList("optimizer_state")
        stateful = optimizer not in _SGD_LIKE
        for t in range(nd):
            shape = (1, sum(p[t] for p in ps), self.tt_ranks[t] * self.tt_q_shapes[t] * self.tt_ranks[t + 1])
            self.tt_cores.append(nn.Parameter(torch.empty(shape, device=device, dtype=torch.float32)))
            self.optimizer_state.append(torch.zeros(shape if stateful else 0, device=device, dtype=torch.float32))
        # every table is initialised as a table of its own cardinality would be
        self.table_ranks = None if table_ranks is None else [[1] + [int(x) for x in rk] + [1] for rk in table_ranks]
        if self.table_ranks is not None:
            assert len(self.table_ranks) == len(Es) and all(len(rk) == nd + 1 and all(a <= b for a, b in zip(rk, self.tt_ranks))
                                                             for rk in self.table_ranks), "table_ranks: per table, <= tt_ranks"
        for k, e in enumerate(Es):
            rk = self.tt_ranks if self.table_ranks is None else self.table_ranks[k]
            one = TableBatchedTTEmbeddingBag.__new__(TableBatchedTTEmbeddingBag)
            nn.Module.__init__(one)
            qk = self.tt_q_shapes if self.table_q is None else self.table_q[k]
            one.num_tables, one.tt_ndim, one.num_embeddings, one.embedding_dim = 1, nd, e, self.out_dim
            one.tt_ranks, one.tt_p_shapes, one.tt_q_shapes = rk, ps[k], qk
            one.tt_cores = [torch.empty((1, ps[k][t], rk[t] * qk[t] * rk[t + 1]), device=device) for t in range(nd)]
            TableBatchedTTEmbeddingBag.reset_parameters(one, weight_dist)
            with torch.no_grad():
                for t in range(nd):
                    self.set_table_core(k, t, one.tt_cores[t][0])
        if self.table_q is not None:  # column k, j: where value j of table k's output row sits in its padded row
            cols = []
            for q in self.table_q:
                at = np.zeros(q, dtype=np.int64)
                for t in range(nd):
                    shape = [1] * nd
                    shape[t] = q[t]
                    at += np.arange(q[t], dtype=np.int64).reshape(shape) * int(np.prod(self.tt_q_shapes[t + 1:]))
                cols.append(at.reshape(-1))
            self.register_buffer("_cols", torch.from_numpy(np.stack(cols)).to(device), persistent=False)
        self.use_cache = False
        self.register_buffer("hashtbl", torch.empty(0, device=device, dtype=torch.int64))
        self.register_buffer("cache_freq", torch.empty(0, device=device, dtype=torch.int64))
        self.register_buffer("cache_state", torch.empty(0, device=device, dtype=torch.int32))
        self.cache_optimizer_state = None
        self.cache_weight = None
        self.warmup = True

    def _table_dims(self, k: int, t: int):
        rk = self.tt_ranks if self.table_ranks is None else self.table_ranks[k]
        return rk[t], (self.tt_q_shapes if self.table_q is None else self.table_q[k])[t], rk[t + 1]

    def set_table_core(self, k: int, t: int, core: torch.Tensor) -> None:
        """core t of table k from its natural shape [p_k_t, r_t q_t r_{t+1}] (the table's own ranks and factoring), zero-padded to
        the common ones"""
        rows = self.table_rows(t)[k]
        if self.table_ranks is None and self.table_q is None:
            rows.copy_(core.reshape(rows.shape))
            return
        R0, Q, R1 = self.tt_ranks[t], self.tt_q_shapes[t], self.tt_ranks[t + 1]
        r0, q, r1 = self._table_dims(k, t)
        rows.zero_()
        rows.view(-1, R0, Q, R1)[:, :r0, :q, :r1] = core.reshape(-1, r0, q, r1)

    def table_core(self, k: int, t: int) -> torch.Tensor:
        """core t of table k in its natural shape [p_k_t, r_t q_t r_{t+1}] (a copy when the table's ranks / factoring are padded)"""
        rows = self.table_rows(t)[k]
        if self.table_ranks is None and self.table_q is None:
            return rows
        R0, Q, R1 = self.tt_ranks[t], self.tt_q_shapes[t], self.tt_ranks[t + 1]
        r0, q, r1 = self._table_dims(k, t)
        return rows.view(-1, R0, Q, R1)[:, :r0, :q, :r1].reshape(rows.shape[0], -1)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, warmup: bool = True,
                per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        res = super().forward(indices, offsets, warmup, per_sample_weights)
        if self.table_q is None:
            return res
        return torch.gather(res, 2, self._cols.unsqueeze(1).expand(-1, res.size(1), -1))  # [tables, B, out_dim]

    def table_rows(self, t: int) -> List[torch.Tensor]:
        """views [p_k_t, slice] of core t, one per table"""
        sizes = [p[t] for p in self.tt_p_shapes]
        return list(torch.split(self.tt_cores[t].detach()[0], sizes, dim=0))

    def full_weight(self) -> torch.Tensor:
        raise NotImplementedError("full_weight() is per table: build a TTEmbeddingBag from table_rows()")


def _lookup_node(fn):
    """the autograd node that runs a group's backward kernels: `fn` itself, or the first node behind the view / reshape nodes
    the module put on top of it (a node with more than one differentiable input, or none behind it, ends the walk)"""
    seen = 0
    while fn is not None and seen < 8:
        name = type(fn).__name__
        if not any(v in name for v in ("View", "Reshape", "Slice", "Gather", "Expand", "Alias", "Unsqueeze", "Squeeze", "Permute", "Transpose", "Clone", "Contiguous")):
            return fn
        nxt = [f for f, _ in fn.next_functions if f is not None]
        if len(nxt) != 1:
            return fn
        fn, seen = nxt[0], seen + 1
    return fn


class MixedTTEmbeddingBag(nn.Module):
    """TT embedding bags for tables of different cardinality -- and, table by table, different TT ranks and different
    factorings q of the (common) embedding dimension.

    Tables that can share a launch set share one module: `self.groups[k]` is the module of group k,
    `self.group_tables[k]` its table ids.
      fused=False: a group = tables of equal (p, q, ranks), a `TableBatchedTTEmbeddingBag`;
      fused=True : a group = tables of equal factoring q whatever their row factors p AND their ranks, a `VarTableTTEmbeddingBag`:
                   ONE launch set per q -- tables of smaller ranks are stored zero-padded to the group's largest (exact: the padding
                   stays zero under the fused optimizers; costs the multiply-adds on the zeros: done by default when that is at most
                   twice the tables' own work, `pad_ranks=True / False` forces it / keeps one group per (q, ranks)).  Tables that differ in q produce
                   output rows of different layouts: separate launch sets, or -- `pad_q` (round 4; same default rule and switch) -- ONE
                   set over the entry-by-entry largest factoring, every table's cores zero-padded to it and its own values gathered
                   out of the padded rows (`VarTableTTEmbeddingBag(table_q=)`).
    `streams=True` gives every group a HIP stream of its own: eagerly the step is host-bound and nothing is gained, but
    captured into a hipGraph (ttx_graph.GraphedRound) the groups become parallel branches and their kernels -- each too
    small to fill the chip at DLRM batch sizes -- run side by side (scripts/bench_mixed.py).
    `tt_ranks` / `tt_q_shapes`: one list for all tables, or one list per table.
    forward(indices, offsets[, per_sample_weights]) takes one tensor per table (nn.EmbeddingBag call form,
    `include_last_offset` as given to the constructor) and returns one [B, D] tensor per table."""

    def __init__(self, num_embeddings: Sequence[int], embedding_dim: int, tt_ranks,
                 tt_p_shapes: Optional[Sequence[Optional[List[int]]]] = None, tt_q_shapes=None,
                 optimizer: OptimType = OptimType.SGD, learning_rate: float = 0.1, eps: float = 1.0e-10,
                 sparse: bool = True, weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False,
                 device: Optional[torch.device] = None, include_last_offset: bool = False,
                 streams: bool = False, fused: bool = False, pad_ranks: Optional[bool] = None,
                 pad_q: Optional[bool] = None) -> None:
        super().__init__()
        self.num_embeddings = [int(e) for e in num_embeddings]
        n = len(self.num_embeddings)
        self.embedding_dim = int(embedding_dim)
        self.include_last_offset = bool(include_last_offset)
        self._streams = None
        per_table = lambda v: v is not None and len(v) > 0 and isinstance(v[0], (list, tuple))  # noqa: E731
        ranks = [[int(x) for x in r] for r in tt_ranks] if per_table(tt_ranks) else [[int(x) for x in tt_ranks]] * n
        assert len(ranks) == n, "tt_ranks: one list, or one list per table"
        if per_table(tt_q_shapes):
            qs = [[int(x) for x in q] for q in tt_q_shapes]
        else:
            qs = [None if tt_q_shapes is None else [int(x) for x in tt_q_shapes]] * n
        assert len(qs) == n, "tt_q_shapes: one list, or one list per table"
        shapes: List[Tuple[int, ...]] = []
        for k, e in enumerate(self.num_embeddings):
            nd = len(ranks[k]) + 1
            given = tt_p_shapes[k] if tt_p_shapes is not None else None
            shapes.append(tuple(int(x) for x in given) if given is not None else tuple(suggested_tt_shapes(e, nd)))
        if fused:  # (every table's factoring, spelled out: the grouping below compares them)
            qs = [q if q is not None else suggested_tt_shapes(self.embedding_dim, len(ranks[k]) + 1, allow_round_up=not enforce_embedding_dim)
                  for k, q in enumerate(qs)]

        def madds(rk, q):
            return sum(rk[i] * q[i] * rk[i + 1] for i in range(len(q)))

        # tables of different FACTORINGS in one launch set (cores zero-padded to the entry-by-entry largest factoring,
        # `VarTableTTEmbeddingBag(table_q=)`, ranks padded with them): decided per group of tables with the same number of cores
        # (round 4 advisor: one global flag q- and rank-padded EVERY group as soon as one passed the test -- possibly onto a
        # factoring outside the specialised templates).  padq_nd = the core counts whose tables share one padded set.
        padq_nd = set()
        if pad_q is None and fused:
            # auto: when the padding costs at most twice the tables' own multiply-adds
            by_nd: Dict[int, List[int]] = {}
            for k in range(n):
                by_nd.setdefault(len(ranks[k]), []).append(k)
            for tabs in by_nd.values():
                if len({tuple(qs[k]) for k in tabs}) < 2 or pad_ranks is False and len({tuple(ranks[k]) for k in tabs}) > 1:
                    continue
                nd1 = len(ranks[tabs[0]])
                qmax = [max(qs[k][i] for k in tabs) for i in range(nd1 + 1)]
                rmax = [max(ranks[k][i] for k in tabs) for i in range(nd1)]
                real = sum(madds([1] + ranks[k] + [1], qs[k]) for k in tabs)
                if madds([1] + rmax + [1], qmax) * len(tabs) <= 2 * real:
                    padq_nd.add(nd1)
        elif pad_q and fused:
            padq_nd = {len(r) for r in ranks}
        pad_q = bool(padq_nd)
        if pad_ranks is None and fused:
            # auto: one launch set per factoring when the zero padding costs at most twice the tables' own multiply-adds
            # (measured, scripts/bench_mixed.py: ranks 32 / 16 in one set 0.278 vs 0.297 ms/step in two; ranks 64 / 32 / 16 /
            # [13,12] in one set 0.72 vs 0.43 in four)
            pad_ranks = True
            by_q: Dict[tuple, List[int]] = {}
            for k in range(n):
                if len(ranks[k]) in padq_nd:  # (q-padded groups pad their ranks anyway: they do not vote on the others' rule)
                    continue
                by_q.setdefault((len(ranks[k]), None if qs[k] is None else tuple(qs[k])), []).append(k)
            for tabs in by_q.values():
                qq = qs[tabs[0]] or suggested_tt_shapes(self.embedding_dim, len(ranks[tabs[0]]) + 1, allow_round_up=not enforce_embedding_dim)
                rmax = [max(ranks[k][i] for k in tabs) for i in range(len(ranks[tabs[0]]))]
                real = sum(madds([1] + ranks[k] + [1], qq) for k in tabs)
                if madds([1] + rmax + [1], qq) * len(tabs) > 2 * real:
                    pad_ranks = False
        groups: Dict[tuple, List[int]] = {}
        for k in range(n):
            # fused: tables of one factoring q share a batched lookup whatever their ranks (smaller ranks are zero-padded to the
            # group's largest, VarTableTTEmbeddingBag(table_ranks=)); pad_ranks=False keeps one group per (q, ranks)
            inq = len(ranks[k]) in padq_nd  # this table's core-count group shares ONE padded factoring (ranks padded with it)
            key = ((len(ranks[k]),) if fused and (pad_ranks or inq) else (tuple(ranks[k]),)) + \
                  ((len(ranks[k]),) if inq else (None if qs[k] is None else tuple(qs[k]),)) + (() if fused else (shapes[k],))
            groups.setdefault(key, []).append(k)
        self.group_tables = list(groups.values())
        self.groups = nn.ModuleList()
        for tables in self.group_tables:
            k0 = tables[0]
            if fused:  # ONE batched lookup for the group's tables, whatever their row factors (VarTableTTEmbeddingBag)
                rmax = [max(ranks[k][i] for k in tables) for i in range(len(ranks[k0]))]
                mixed_ranks = any(ranks[k] != rmax for k in tables)
                qmax = [max(qs[k][i] for k in tables) for i in range(len(qs[k0]))]
                mixed_q = any(qs[k] != qmax for k in tables)
                self.groups.append(VarTableTTEmbeddingBag(
                    [self.num_embeddings[k] for k in tables], self.embedding_dim, rmax, [list(shapes[k]) for k in tables],
                    qmax, optimizer, learning_rate, eps, sparse, weight_dist, enforce_embedding_dim, device, True,
                    [ranks[k] for k in tables] if mixed_ranks else None, [qs[k] for k in tables] if mixed_q else None))
            else:
                self.groups.append(TableBatchedTTEmbeddingBag(
                    len(tables), max(self.num_embeddings[k] for k in tables), self.embedding_dim, list(ranks[k0]),
                    list(shapes[k0]), qs[k0], optimizer, learning_rate, eps, sparse, False, 0, 0, weight_dist,
                    enforce_embedding_dim, device, True))
        self._streams = [torch.cuda.Stream(device=self.groups[0].tt_cores[0].device) for _ in self.groups] \
            if streams and len(self.groups) > 1 else None

    def forward(self, indices: Sequence[torch.Tensor], offsets: Sequence[torch.Tensor],
                per_sample_weights: Optional[Sequence[Optional[torch.Tensor]]] = None) -> List[torch.Tensor]:
        n = len(self.num_embeddings)
        assert len(indices) == n and len(offsets) == n, f"one (indices, offsets) pair per table: {n} tables"
        outs: List[Optional[torch.Tensor]] = [None] * n
        cur = torch.cuda.current_stream() if self._streams else None
        for g, (mod, tables) in enumerate(zip(self.groups, self.group_tables)):
            idx, off = merge_bags([indices[k] for k in tables], [offsets[k] for k in tables], self.include_last_offset)
            psw = None
            if per_sample_weights is not None and any(per_sample_weights[k] is not None for k in tables):
                psw = torch.cat([per_sample_weights[k].reshape(-1) if per_sample_weights[k] is not None
                                 else torch.ones(indices[k].numel(), device=idx.device) for k in tables])
            if self._streams:
                s = self._streams[g]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    res = mod(idx, off, True, psw)  # [tables, B, D]
                node = _lookup_node(res.grad_fn)
                if node is not None:
                    # autograd runs this group's backward (recompute + fused optimizer) on the group's stream as well, and
                    # with a fused optimizer there is no leaf gradient whose stream the engine would join at the end: join
                    # it here, after the node has enqueued its kernels (needed for hipGraph capture -- "unjoined work" --
                    # and for whoever reads the cores on the caller's stream next).  The hook goes on the LOOKUP node, not on
                    # whatever reshapes its result (q0 > 4 returns a view: a ViewBackward hook would fire before the lookup's
                    # backward has enqueued anything -- round 3 advisor finding)
                    node.register_hook(lambda gi, go, s=s, cur=cur: cur.wait_stream(s))
                for t in (idx, off) + ((psw,) if psw is not None else ()):
                    t.record_stream(s)   # allocated on the caller's stream, read on the group's
                res.record_stream(cur)   # ... and the other way round
            else:
                res = mod(idx, off, True, psw)
            for j, k in enumerate(tables):
                outs[k] = res[j]
        if self._streams:
            for s in self._streams:
                cur.wait_stream(s)
        return outs  # type: ignore[return-value]
