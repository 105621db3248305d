"""Table-sharded multi-GPU TT embedding lookup (one process per GPU).

The reference is single-device.  Tables are
independent, so the path shards by table: table t lives on rank t % W with its
TT cores, optimizer state and fused optimizer entirely rank-local (no gradient
all-reduce -- the cores are model-parallel).  Per step there are exactly two
exchanges, both `all_to_all_single` over RCCL/xGMI (point-to-point links, every
pair of GPUs talks directly):

  1. lookups in : every rank sends, to each owner, the (lengths, indices) of its
                  LOCAL batch for the owner's tables;
  2. pooled out : every owner returns the pooled [tables_owned, B_local, D] block
                  of each requester's batch; backward sends the gradient of that
                  block the opposite way.

forward(indices, offsets) takes the local batch for ALL tables (table-major,
include_last_offset form: offsets has num_tables*B_local + 1 entries) and
returns [num_tables, B_local, D], like TableBatchedTTEmbeddingBag on one GPU.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn

import tt_embeddings_ops as _ops

# test hook: run the exchange code path even with a single rank (bench.py --force-sharded on a 1-GPU box)
_COMPACT_RAGGED = os.environ.get("TTX_PADDED_WEIGHTS", "") != "1"  # (A/B: "1" = round 4's zero-weight padding)
_FORCE_EXCHANGE = bool(os.environ.get("TTX_FORCE_EXCHANGE"))


def _a2a(group, out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits) -> None:
    """all_to_all_single straight on the process group's backend object: the python wrapper of
    torch.distributed spends ~30 us per call on argument checking, more than the exchange itself takes at
    the benchmark's message sizes.  Equal splits go without split lists (the backend's fast path)."""
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    if out_splits is not None and len(set(out_splits)) <= 1 and in_splits is not None and len(set(in_splits)) <= 1:
        out_splits, in_splits = [], []
    try:
        pg.alltoall_base(out, inp, out_splits or [], in_splits or []).wait()
    except (AttributeError, TypeError):  # an older / different backend object: the public entry point
        dist.all_to_all_single(out, inp, out_splits or None, in_splits or None, group=group)


class DirectExchange:
    """All-to-all straight on RCCL (csrc/ttx_torch.cpp), on the CURRENT stream: no side stream, no event hops,
    ~5 us per exchange instead of ~17 + wrapper time, and capturable in a hipGraph (torch.distributed's
    collectives are not: its watchdog aborts on an event recorded in a capturing stream).  Equal splits go through
    ncclAllToAll, per-peer counts (tables not a multiple of the world size: 26 on 8 ranks) through one
    ncclSend/ncclRecv group.  One communicator per process group, bootstrapped through torch.distributed.
    close() destroys the communicator (ncclCommDestroy was seen to hang on a one-rank communicator in this pool's boxes:
    bench.py calls it under a timeout); otherwise it dies with the process."""

    def __init__(self, group, device: torch.device) -> None:
        import ttx_torch

        self._lib = ttx_torch
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        box = [ttx_torch.rccl_unique_id() if self.rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        index = device.index if device.index is not None else torch.cuda.current_device()
        self.comm = ttx_torch.rccl_comm_init(box[0], self.rank, self.world, index)

    def close(self) -> None:
        """ncclCommDestroy (collective).  Callers that must not hang run it under a timeout (bench.py): it was seen not
        to return on a one-rank communicator in a sandboxed box."""
        comm, self.comm = self.comm, None
        if comm is not None:
            torch.cuda.synchronize()
            self._lib.rccl_comm_destroy(comm)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None) -> None:
        """block p of `inp` (in_splits[p] elements) -> rank p; block p of `out` (out_splits[p] elements) <- rank p.
        Splits are element counts; None = equal blocks."""
        if _equal_splits(out_splits, in_splits):
            self._lib.rccl_all_to_all(self.comm, out, inp, self.world)
        else:
            self._lib.rccl_all_to_allv(self.comm, out, inp, [int(x) for x in in_splits], [int(x) for x in out_splits])


class CollectiveExchange:
    """The DirectExchange interface on top of torch.distributed.all_to_all_single (any backend).  What the tests
    use to drive the direct route's split bookkeeping on CPU (gloo); never faster than the plain route."""

    def __init__(self, group) -> None:
        self.group = group
        self.world = dist.get_world_size(group)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None) -> None:
        if _equal_splits(out_splits, in_splits):
            out_splits = in_splits = None
        dist.all_to_all_single(out.view(-1), inp.contiguous().view(-1), None if out_splits is None else [int(x) for x in out_splits],
                               None if in_splits is None else [int(x) for x in in_splits], group=self.group)


def _equal_splits(out_splits, in_splits) -> bool:
    return (out_splits is None or len(set(out_splits)) <= 1) and (in_splits is None or len(set(in_splits)) <= 1)


class _DirectPooledAllToAll(torch.autograd.Function):
    """differentiable all-to-all of row blocks ([rows, D], split by rows) through a DirectExchange"""

    @staticmethod
    def forward(ctx, ex, x: torch.Tensor, in_rows: List[int], out_rows: List[int]) -> torch.Tensor:
        ctx.ex, ctx.in_rows, ctx.out_rows = ex, in_rows, out_rows
        x = x.contiguous()
        D = x.shape[1]
        out = x.new_empty((sum(out_rows), D))
        ex.all_to_all(out, x, [r * D for r in out_rows], [r * D for r in in_rows])
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        g = g.contiguous()
        D = g.shape[1]
        gin = g.new_empty((sum(ctx.in_rows), D))
        ctx.ex.all_to_all(gin, g, [r * D for r in ctx.in_rows], [r * D for r in ctx.out_rows])
        return None, gin, None, None


class _PooledAllToAll(torch.autograd.Function):
    """differentiable all_to_all_single with explicit split sizes (rows of D floats)"""

    @staticmethod
    def forward(ctx, group, x: torch.Tensor, in_splits: List[int], out_splits: List[int]) -> torch.Tensor:
        ctx.group, ctx.in_splits, ctx.out_splits = group, in_splits, out_splits
        out = x.new_empty((sum(out_splits),) + tuple(x.shape[1:]))
        _a2a(group, out, x.contiguous(), out_splits, in_splits)
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        gin = g.new_empty((sum(ctx.in_splits),) + tuple(g.shape[1:]))
        _a2a(ctx.group, gin, g.contiguous(), ctx.in_splits, ctx.out_splits)
        return None, gin, None, None


class ShardedTableBatchedTTEmbeddingBag(nn.Module):
    """`num_tables` identical-shape TT tables sharded table-wise over a process group.

    Constructor keywords are those of TableBatchedTTEmbeddingBag plus `group`.
    `fixed_pooling=L` (optional, forward kw) promises every bag holds exactly L
    lookups, which removes the host read-back of split sizes."""

    def __init__(self, num_tables: int, num_embeddings: int, embedding_dim: int, tt_ranks: List[int],
                 group: Optional["dist.ProcessGroup"] = None, index_wire_dtype: Optional[torch.dtype] = None, **kw) -> None:
        super().__init__()
        self.group = group
        # (round 6) what the "lookups in" exchange carries per index: int32 whenever the tables' row count allows it (every
        # BASELINE config: E = 11M) -- half the bytes of the reference's int64 indices on the xGMI links (SURVEY.md section 8d:
        # 8 B of the 8 + 4 D / L bytes a lookup costs on the wire at D = 64, L = 20 are the index).  The owner widens what it
        # received on the device; lengths of ragged bags travel as int32 as well.  `index_wire_dtype=torch.int64` keeps round 5's wire.
        if index_wire_dtype is None:
            index_wire_dtype = torch.int32 if int(num_embeddings) < (1 << 31) else torch.int64
        if index_wire_dtype not in (torch.int32, torch.int64) or (index_wire_dtype == torch.int32 and int(num_embeddings) >= (1 << 31)):
            raise ValueError("index_wire_dtype: torch.int32 (tables of fewer than 2^31 rows) or torch.int64")
        self.index_wire_dtype = index_wire_dtype
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.num_tables, self.embedding_dim = num_tables, embedding_dim
        W = self.world
        self.owned = [[t for t in range(num_tables) if t % W == d] for d in range(W)]
        self.my_tables = self.owned[self.rank]
        # owner-major table order used on the wire, and its inverse
        order = [t for d in range(W) for t in self.owned[d]]
        inv = [0] * num_tables
        for pos, t in enumerate(order):
            inv[t] = pos
        self._order, self._inv = order, inv
        self._identity = order == list(range(num_tables))  # e.g. one table per rank: no reordering on the wire
        self.direct: Optional[DirectExchange] = None  # enable_direct_exchange()
        self._dev_cache = {}  # device tensors that only depend on the batch shape (built once, not per step)
        assert not kw.get("use_cache", False), "cache is single-table only (reference :458)"
        self.local = None
        if self.my_tables:
            self.local = _ops.TableBatchedTTEmbeddingBag(len(self.my_tables), num_embeddings, embedding_dim, tt_ranks, **kw)

    def enable_direct_exchange(self, exchange=None) -> None:
        """Route the fixed-pooling exchanges through RCCL directly (see DirectExchange): any number of tables per
        rank (uneven ownership goes through a ncclSend/ncclRecv group).  Needs the C++ extension; collective call --
        every rank must make it.  `exchange`: an object with DirectExchange's all_to_all (tests: CollectiveExchange)."""
        if exchange is not None:
            self.direct = exchange
        elif self.direct is None:
            dev = (next(self.local.parameters()).device if self.local is not None
                   else torch.device("cuda", torch.cuda.current_device()))
            self.direct = DirectExchange(self.group, dev)

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        _a2a(self.group, out, inp, out_splits, in_splits)

    def prefetch_many(self, batches, fixed_pooling: Optional[int] = None) -> bool:
        """Plan a round of batches ahead (collective: every rank calls it with its batches of the same round).  The
        lookups' way in -- which depends on the batches' indices only -- is done for ALL of them at once: ONE index
        exchange carries the round (K times the bytes of a step's exchange, which is latency-bound), one copy brings it
        into table-major order, and the owners' lookup prologues run in one launch (`local.prefetch_many`).  Each batch's
        next `forward(indices, offsets, fixed_pooling=L)` -- same tensor objects, unmodified -- then starts at the local
        lookup: two exchanges per step instead of three.  Fixed pooling only (the split sizes are known without a host
        read-back); returns False, and does nothing, otherwise."""
        batches = list(batches)
        W, NT = self.world, self.num_tables
        if fixed_pooling is None or not batches or (W == 1 and not _FORCE_EXCHANGE):
            return False
        Lp, K = int(fixed_pooling), len(batches)
        B = (batches[0][1].numel() - 1) // NT
        for i, o in batches:
            if i.dim() != 1 or o.dim() != 1 or i.numel() != NT * B * Lp or (o.numel() - 1) // NT != B:
                return False
        dev = batches[0][0].device
        n_own = [len(o) for o in self.owned]
        n_me = n_own[self.rank]
        order = self._cached(("order", dev), lambda: torch.tensor(self._order, device=dev))
        # send: [table (owner-major)][batch][B * Lp] -- the block of destination d is its tables' rows, contiguous
        stack = torch.stack([i.to(self.index_wire_dtype).view(NT, B * Lp) for i, _ in batches], dim=1)   # [NT, K, B*Lp]
        send_idx = (stack if self._identity else stack[order]).contiguous().view(-1)
        in_splits = [k * K * B * Lp for k in n_own]
        out_splits = [n_me * K * B * Lp] * W
        recv_idx = send_idx.new_empty(sum(out_splits))
        if self.direct is not None:
            self.direct.all_to_all(recv_idx, send_idx, out_splits, in_splits)
        else:
            self._a2a(recv_idx, send_idx, out_splits, in_splits)
        # wire order [src][k][batch][b][l] -> per batch table-major [k][src][b][l]
        loc = recv_idx.view(W, n_me, K, B * Lp).permute(2, 1, 0, 3).long().contiguous().view(K, -1)  # (widened with the reorder)
        loc_off = self._cached(("off", dev, n_me * W * B, Lp), lambda: torch.arange(
            0, n_me * W * B * Lp + 1, Lp, device=dev, dtype=torch.int64))
        loc_idx = [loc[j] for j in range(K)]
        if n_me and hasattr(self.local, "prefetch_many"):
            self.local.prefetch_many([(x, loc_off) for x in loc_idx])  # (False on CPU tensors: prologues then run in line)
        if not hasattr(self, "_planned"):
            self._planned = {}
        if len(self._planned) + K > 64:
            raise RuntimeError(f"ShardedTableBatchedTTEmbeddingBag.prefetch_many: {len(self._planned)} planned batches are still "
                               f"pending; consume them with forward() or call drop_planned() on every rank first")
        for j, (i, o) in enumerate(batches):  # keyed by the caller's OWN tensor objects (forward looks them up before any cast)
            self._planned[(id(i), id(o))] = (i, o, i._version, o._version, Lp, loc_idx[j], loc_off)
        return True

    def drop_planned(self) -> None:
        """Forget the batches planned ahead and not yet consumed (every rank must do the same: whether a step's index
        exchange runs is decided per rank from this table)."""
        if getattr(self, "_planned", None):
            self._planned.clear()
        if self.local is not None and hasattr(self.local, "_drop_prefetched"):
            self.local._drop_prefetched()  # (through the module's own bookkeeping: a cache-using local module counts each batch once)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, fixed_pooling: Optional[int] = None,
                max_pooling: Optional[int] = None) -> torch.Tensor:
        """`fixed_pooling=L`: every bag holds exactly L lookups -- the exchanges' split sizes are known without a host
        read-back (and the step is capturable / can be planned ahead).  `max_pooling=L` (round 4): RAGGED bags of at most L
        lookups each take the same route -- every bag travels padded to L lookups, the bags' lengths beside the indices (a
        fixed-size exchange of B ints per table): fixed shapes everywhere, no `.tolist()`, capturable.  Round 5: the OWNER
        compacts what it received on the device and hands its local lookup the live count as a device scalar
        (`forward(n_dev=)`, ttx_lookup_prologue_n) -- only the real lookups are planned and contracted (round 4 contracted
        the padding with weight zero: L / mean length times the lookups).  A bag longer than L is TRUNCATED -- the caller's contract, like
        fixed_pooling's; `TTX_CHECK_POOLING=1` verifies it with a host read-back.  Neither: ragged bags with one host
        read-back of the split sizes per step."""
        W, NT, D = self.world, self.num_tables, self.embedding_dim
        if max_pooling is not None and fixed_pooling is None:
            # (round 4 advisor) the SAME semantics at every world size -- one rank included: bags longer than L are truncated, the
            # padding is weighted zero -- and the same guard as below: batches planned ahead and not consumed are an error here
            # too (a rank that takes this route while its peers consume a planned batch would issue other collectives)
            if getattr(self, "_planned", None):
                raise RuntimeError("ShardedTableBatchedTTEmbeddingBag.forward(max_pooling=..): batches planned ahead "
                                   "(prefetch_many) are pending; consume them or call drop_planned() on every rank")
            return self._forward_padded(indices.long(), offsets.long(), int(max_pooling))
        # a batch planned ahead is found by the caller's own tensor objects -- before the casts below replace them
        planned = getattr(self, "_planned", None)
        hit = planned.pop((id(indices), id(offsets)), None) if planned else None
        if planned is not None and hit is None and len(planned) > 0:
            # Whether this step's index exchange runs is decided by every rank on its own: a rank that misses while its
            # peers hit would issue a collective the others skip, and the round would hang.  So a miss is an error.
            raise RuntimeError("ShardedTableBatchedTTEmbeddingBag.forward: batches planned ahead (prefetch_many) are pending "
                               "and this batch is not one of them; pass the planned tensor objects, or call drop_planned() "
                               "on every rank")
        if hit is not None and not (hit[0] is indices and hit[1] is offsets and hit[2] == indices._version
                                    and hit[3] == offsets._version and fixed_pooling is not None and hit[4] == int(fixed_pooling)):
            raise RuntimeError("ShardedTableBatchedTTEmbeddingBag.forward: this batch was planned ahead (prefetch_many) and "
                               "has been written to, or is used with another fixed_pooling, since; falling back to an in-line "
                               "index exchange on this rank alone would hang the other ranks")
        indices, offsets = indices.long(), offsets.long()
        B = (offsets.numel() - 1) // NT
        if W == 1 and not _FORCE_EXCHANGE:
            return self.local(indices, offsets)
        dev = indices.device
        n_own = [len(o) for o in self.owned]
        n_me = n_own[self.rank]
        order = self._cached(("order", dev), lambda: torch.tensor(self._order, device=dev))
        # ---- 1. lookups in -------------------------------------------------
        if hit is not None:  # planned ahead (prefetch_many): the index exchange of this batch is done
            loc_idx, loc_off = hit[5], hit[6]
        elif fixed_pooling is not None:
            Lp = int(fixed_pooling)
            wire = indices.to(self.index_wire_dtype)
            send_idx = wire if self._identity else wire.view(NT, B * Lp)[order].contiguous().view(-1)
            in_splits = [k * B * Lp for k in n_own]
            out_splits = [n_me * B * Lp] * W
            recv_idx = wire.new_empty(sum(out_splits))
            if self.direct is not None:
                self.direct.all_to_all(recv_idx, send_idx.contiguous(), out_splits, in_splits)
            else:
                self._a2a(recv_idx, send_idx, out_splits, in_splits)
            # wire order [src][k][b][l] -> table-major [k][src][b][l]
            loc_idx = recv_idx.view(W, n_me, B * Lp).permute(1, 0, 2).long().contiguous().view(-1)
            loc_off = self._cached(("off", dev, n_me * W * B, Lp), lambda: torch.arange(
                0, n_me * W * B * Lp + 1, Lp, device=dev, dtype=torch.int64))
        else:
            lengths = (offsets[1:] - offsets[:-1]).view(NT, B)
            send_len = lengths[order].contiguous()                       # owner-major
            recv_len = lengths.new_empty((W, n_me, B))
            self._a2a(recv_len.view(-1), send_len.view(-1), [n_me * B] * W, [k * B for k in n_own])
            # split sizes of the index exchange: ONE host read-back
            tbl_cnt = send_len.sum(dim=1)
            bounds = [0]
            for k in n_own:
                bounds.append(bounds[-1] + k)
            host = torch.cat([torch.stack([tbl_cnt[bounds[d]:bounds[d + 1]].sum() for d in range(W)]),
                              recv_len.view(W, -1).sum(dim=1)]).tolist()
            in_splits, out_splits = [int(x) for x in host[:W]], [int(x) for x in host[W:]]
            # indices in owner-major table order
            tbl_off = offsets[::B]                                         # [NT+1] start of each table's run
            seg_start = tbl_off[:-1][order]
            seg_len = tbl_cnt
            send_idx = indices[_segment_gather(seg_start, seg_len, sum(in_splits))].to(self.index_wire_dtype)
            recv_idx = send_idx.new_empty(sum(out_splits))
            self._a2a(recv_idx, send_idx, out_splits, in_splits)
            recv_idx = recv_idx.long()
            # per-(src, k) segments -> table-major [k][src]
            seg_len_w = recv_len.sum(dim=2)                                # [W, n_me] wire order
            seg_start_w = (torch.cumsum(seg_len_w.view(-1), 0) - seg_len_w.view(-1)).view(W, n_me)
            loc_idx = recv_idx[_segment_gather(seg_start_w.t().reshape(-1), seg_len_w.t().reshape(-1), sum(out_splits))]
            loc_len = recv_len.permute(1, 0, 2).reshape(-1)                # [k][src][b]
            loc_off = torch.cat([loc_len.new_zeros(1), torch.cumsum(loc_len, 0)])
        # ---- 2. local lookup of the GLOBAL batch for my tables ---------------
        if n_me:
            pooled = self.local(loc_idx, loc_off)                          # [n_me, W*B, D]
            send = pooled.view(n_me, W, B, D).permute(1, 0, 2, 3).reshape(W * n_me * B, D)
        else:
            # (a rank that owns no table still takes part in BOTH exchanges: without requires_grad its autograd
            #  node would be missing, it would skip the backward all-to-all and the owners would wait for ever)
            send = torch.zeros((0, D), device=dev, dtype=torch.float32, requires_grad=True)
        # ---- 3. pooled out ---------------------------------------------------
        if self.direct is not None and fixed_pooling is not None:
            got = _DirectPooledAllToAll.apply(self.direct, send, [n_me * B] * W, [k * B for k in n_own])
        else:
            got = _PooledAllToAll.apply(self.group, send, [n_me * B] * W, [k * B for k in n_own])
        out = got.view(NT, B, D)
        if self._identity:
            return out
        return out[self._cached(("inv", dev), lambda: torch.tensor(self._inv, device=dev))]

    def _forward_padded(self, indices: torch.Tensor, offsets: torch.Tensor, Lp: int) -> torch.Tensor:
        """ragged bags as fixed bags of Lp lookups, the padding weighted zero (see forward)"""
        W, NT, D = self.world, self.num_tables, self.embedding_dim
        B = (offsets.numel() - 1) // NT
        dev = indices.device
        n_own = [len(o) for o in self.owned]
        n_me = n_own[self.rank]
        lengths = offsets[1:] - offsets[:-1]                                   # [NT * B]
        if os.environ.get("TTX_CHECK_POOLING") and lengths.numel() and int(lengths.max()) > Lp:
            raise ValueError(f"max_pooling={Lp}: a bag holds {int(lengths.max())} lookups")
        pos = self._cached(("pos", dev, Lp), lambda: torch.arange(Lp, device=dev, dtype=torch.int64))
        valid = pos[None, :] < lengths[:, None]                               # [NT * B, Lp]
        if indices.numel():
            src = (offsets[:-1, None] + pos[None, :]).clamp_(max=indices.numel() - 1)
            idx_pad = torch.where(valid, indices[src], indices.new_zeros(()))
        else:
            idx_pad = indices.new_zeros((NT * B, Lp))
        order = self._cached(("order", dev), lambda: torch.tensor(self._order, device=dev))
        idx_pad = idx_pad.to(self.index_wire_dtype)
        send_idx = (idx_pad.view(NT, B * Lp) if self._identity else idx_pad.view(NT, B * Lp)[order]).contiguous().view(-1)
        lengths = lengths.clamp(max=Lp)  # (a bag longer than L is truncated: what travels is what the padded rows hold)
        lengths = lengths.to(self.index_wire_dtype)
        send_len = (lengths.view(NT, B) if self._identity else lengths.view(NT, B)[order]).contiguous().view(-1)
        alone = W == 1 and not _FORCE_EXCHANGE  # one rank: nothing to exchange, the same padded lookup
        if alone:
            recv_idx, recv_len = send_idx, send_len
        else:
            recv_idx = send_idx.new_empty(W * n_me * B * Lp)
            recv_len = lengths.new_empty(W * n_me * B)
            ex = self.direct.all_to_all if self.direct is not None else self._a2a
            ex(recv_idx, send_idx, [n_me * B * Lp] * W, [k * B * Lp for k in n_own])
            ex(recv_len, send_len, [n_me * B] * W, [k * B for k in n_own])
        # wire order [src][k][b] -> table-major [k][src][b]
        loc_idx = recv_idx.view(W, n_me, B * Lp).permute(1, 0, 2).long().contiguous().view(-1)
        loc_len = recv_len.view(W, n_me, B).permute(1, 0, 2).long().reshape(-1)
        if n_me and _COMPACT_RAGGED:
            # (round 5) the owner contracts the REAL lookups only: the padded rows are compacted on the device -- lookup l of bag b
            # goes to off[b] + l, the padding to a dump slot behind the buffer (fixed shapes: scatter, no boolean indexing) -- and
            # the local lookup is told the live count on the device (forward(n_dev=), ttx_lookup_prologue_n).  No host read-back,
            # capturable, and no longer L / mean-length times the lookups (round 4 contracted the padding with weight zero).
            cap = n_me * W * B * Lp
            loc_off = torch.cat([loc_len.new_zeros(1), torch.cumsum(loc_len, 0)])
            dest = torch.where(pos[None, :] < loc_len[:, None], loc_off[:-1, None] + pos[None, :], loc_off.new_full((), cap))
            comp = loc_idx.new_zeros(cap + 1).scatter_(0, dest.view(-1), loc_idx)[:cap]
            pooled = self.local(comp, loc_off, n_dev=loc_off[-1:].to(torch.int32))   # [n_me, W*B, D]
            send = pooled.view(n_me, W, B, D).permute(1, 0, 2, 3).reshape(W * n_me * B, D)
        elif n_me:
            loc_w = (pos[None, :] < loc_len[:, None]).to(torch.float32).view(-1)
            loc_off = self._cached(("off", dev, n_me * W * B, Lp), lambda: torch.arange(
                0, n_me * W * B * Lp + 1, Lp, device=dev, dtype=torch.int64))
            pooled = self.local(loc_idx, loc_off, per_sample_weights=loc_w)   # [n_me, W*B, D]
            send = pooled.view(n_me, W, B, D).permute(1, 0, 2, 3).reshape(W * n_me * B, D)
        else:
            send = torch.zeros((0, D), device=dev, dtype=torch.float32, requires_grad=True)
        if alone:
            got = send
        elif self.direct is not None:
            got = _DirectPooledAllToAll.apply(self.direct, send, [n_me * B] * W, [k * B for k in n_own])
        else:
            got = _PooledAllToAll.apply(self.group, send, [n_me * B] * W, [k * B for k in n_own])
        out = got.view(NT, B, D)
        if self._identity:
            return out
        return out[self._cached(("inv", dev), lambda: torch.tensor(self._inv, device=dev))]

    def _cached(self, key, make):
        v = self._dev_cache.get(key)
        if v is None:
            v = self._dev_cache[key] = make()
        return v

    def set_learning_rate(self, lr: float) -> None:
        if self.local is not None:
            self.local.set_learning_rate(lr)


def _segment_gather(seg_start: torch.Tensor, seg_len: torch.Tensor, total: int) -> torch.Tensor:
    """positions of the concatenation of segments [start_i, start_i + len_i)"""
    dst_start = torch.cumsum(seg_len, 0) - seg_len
    rep = torch.repeat_interleave(seg_start - dst_start, seg_len, output_size=total)
    return rep + torch.arange(total, device=seg_start.device, dtype=seg_start.dtype)
