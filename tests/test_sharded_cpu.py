"""world_size-2 gloo test (CPU, no GPU) of the table-sharded path: routing of
lookups to table owners, pooled vectors back, gradients the opposite way, fused
optimizer rank-local.  The local lookup engine is the oracle (test-only
injection); the reference has no distributed code to compare with, so the check
is against the single-process TableBatchedTTEmbeddingBag on the same tables."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

P, Q, R = [7, 9, 11], [3, 4, 5], [13, 12]
NT, B_LOCAL, D = 5, 6, 60


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_inputs(rank, fixed, NT=NT):
    rs = np.random.RandomState(100 + rank)
    E_ = int(np.prod(P))
    if fixed > 0:
        lengths = np.full(NT * B_LOCAL, fixed, dtype=np.int64)
    else:
        lengths = rs.randint(0, 5, size=NT * B_LOCAL).astype(np.int64)
    idx = rs.randint(0, E_, size=int(lengths.sum())).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    grad = (rs.rand(NT, B_LOCAL, D) * 0.1).astype(np.float32)
    return idx, off, grad


def _worker(rank, world, port, fixed, q, NT=NT, direct=False, wire=None):
    try:
        for p in (HERE, os.path.join(ROOT, "fbtt-embedding_amd")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import gen_inputs as G
        import oracle_engine
        import tt_embeddings_ops as ops
        import ttx_sharded

        ops._engine = oracle_engine  # CPU stand-in for the HIP engine (test only)
        cores_all = G.make_cores(5, NT, P, Q, R)
        extra = {} if wire is None else {"index_wire_dtype": getattr(torch, wire)}
        m = ttx_sharded.ShardedTableBatchedTTEmbeddingBag(
            NT, int(np.prod(P)), D, R, tt_p_shapes=P, tt_q_shapes=Q, sparse=True, optimizer=ops.OptimType.SGD,
            learning_rate=0.1, weight_dist="uniform", device="cpu", **extra)
        mine = m.my_tables
        if direct:  # the direct route's bookkeeping (per-peer element counts), carried by gloo instead of RCCL
            m.enable_direct_exchange(ttx_sharded.CollectiveExchange(None))
        # what integer dtype travels (round 6: int32 indices / lengths on the wire unless asked otherwise)
        seen = set()
        real_a2a = dist.all_to_all_single

        def spy(out, inp, *a, **k):
            if not inp.dtype.is_floating_point:
                seen.add(str(inp.dtype))
            return real_a2a(out, inp, *a, **k)
        dist.all_to_all_single = spy
        real_base = ttx_sharded._a2a

        def spy_base(group, out, inp, out_splits, in_splits):
            if not inp.dtype.is_floating_point:
                seen.add(str(inp.dtype))
            return real_base(group, out, inp, out_splits, in_splits)
        ttx_sharded._a2a = spy_base
        if m.local is not None:
            with torch.no_grad():
                for t, core in enumerate(m.local.tt_cores):
                    core.copy_(torch.from_numpy(cores_all[t][mine]))
        idx, off, grad = _make_inputs(rank, fixed, NT)
        if fixed < 0:  # ragged bags through the fixed-capacity exchange (max_pooling = -fixed; longer bags are truncated)
            out = m(torch.from_numpy(idx), torch.from_numpy(off), max_pooling=-fixed)
        else:
            out = m(torch.from_numpy(idx), torch.from_numpy(off), fixed_pooling=fixed or None)
        out.backward(torch.from_numpy(grad))
        want = {"torch." + (wire or "int32")}
        if fixed == 0:
            want.add("torch.int64")  # (ragged bags with a host read-back: the lengths travel as they are; the indices as `wire`)
        assert seen and seen <= want and ("torch." + (wire or "int32")) in seen, f"integer dtypes on the wire: {seen}, expected {want}"
        q.put((rank, out.detach().numpy(), mine, [c.detach().numpy() for c in m.local.tt_cores] if m.local is not None else []))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, "ERROR", traceback.format_exc(), None))


# NT == world: one table per rank (the bench shape); NT = 5: uneven ownership (3 + 2 tables); NT = 1: a rank that owns
# no table still has to take part in both exchanges, forward and backward; direct: DirectExchange's split lists
# fixed < 0 (round 5): RAGGED bags with max_pooling = -fixed -- padded rows on the wire, compacted by the owner, the local lookup
# told its live count (n_dev); -4 holds every bag (lengths 0..4), -3 truncates the bags of four
@pytest.mark.parametrize("fixed,NT,direct,wire", [(0, 5, False, None), (3, 5, False, None), (3, 2, False, None), (3, 5, True, None),
                                                  (3, 2, True, None), (3, 1, False, None), (0, 1, False, None), (3, 1, True, None),
                                                  (2, 7, True, None), (-4, 5, False, None), (-3, 5, True, None), (-4, 1, False, None),
                                                  (-3, 2, False, None),
                                                  # round 6: int32 on the wire is the default above; the reference's int64 on request
                                                  (3, 5, True, "int64"), (0, 5, False, "int64"), (-3, 5, False, "int64")])
def test_two_rank_table_sharding_matches_single_process(fixed, NT, direct, wire):
    sys.path.insert(0, HERE)
    import gen_inputs as G
    import oracle_lib as O

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fixed, q, NT, direct, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=180)
        assert r[1] is not None and not isinstance(r[1], str), f"rank {r[0]} failed:\n{r[2]}"
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    # single-process expectation: all NT tables, global batch = ranks' batches side by side
    cores = G.make_cores(5, NT, P, Q, R)
    g = O.make_geom(NT, P, Q, R)
    per_rank = [_make_inputs(r, fixed, NT) for r in range(world)]
    # global table-major order: table t -> [rank0 bags | rank1 bags]
    idx_g, len_g = [], []
    for t in range(NT):
        for r in range(world):
            idx, off, _ = per_rank[r]
            if fixed < 0:  # max_pooling: the first -fixed lookups of every bag
                for b in range(t * B_LOCAL, (t + 1) * B_LOCAL):
                    n = min(int(off[b + 1] - off[b]), -fixed)
                    idx_g.append(idx[off[b]:off[b] + n])
                    len_g.append(np.array([n], dtype=np.int64))
                continue
            lo, hi = off[t * B_LOCAL], off[(t + 1) * B_LOCAL]
            idx_g.append(idx[lo:hi])
            len_g.append(np.diff(off[t * B_LOCAL:(t + 1) * B_LOCAL + 1]))
    idx_g = np.concatenate(idx_g)
    off_g = np.concatenate([[0], np.cumsum(np.concatenate(len_g))]).astype(np.int64)
    BG = world * B_LOCAL
    rowidx, tableidx = O.rowidx_from_offsets(off_g, NT)
    out_g = O.tt_forward(g, BG, D, idx_g, rowidx, tableidx, cores)
    grad_g = np.concatenate([per_rank[r][2] for r in range(world)], axis=1)  # [NT, BG, D]
    new_cores = [c.copy() for c in cores]
    O.tt_backward(g, O.OPTIM_SGD, BG, D, 0.1, 0.0, idx_g, rowidx, tableidx, grad_g, new_cores)
    for r in range(world):
        _, out, mine, cr = res[r]
        np.testing.assert_allclose(out, out_g[:, r * B_LOCAL:(r + 1) * B_LOCAL], rtol=1e-5, atol=1e-7)
        assert mine == [t for t in range(NT) if t % world == r]
        for t in range(3 if mine else 0):
            np.testing.assert_allclose(cr[t], new_cores[t][mine], rtol=1e-5, atol=1e-7)


def _worker_round(rank, world, port, q, NT, direct):
    try:
        for p in (HERE, os.path.join(ROOT, "fbtt-embedding_amd")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import gen_inputs as G
        import oracle_engine
        import tt_embeddings_ops as ops
        import ttx_sharded

        ops._engine = oracle_engine
        cores_all = G.make_cores(5, NT, P, Q, R)
        fixed, K = 3, 3

        def fresh():
            m = ttx_sharded.ShardedTableBatchedTTEmbeddingBag(
                NT, int(np.prod(P)), D, R, tt_p_shapes=P, tt_q_shapes=Q, sparse=True, optimizer=ops.OptimType.SGD,
                learning_rate=0.1, weight_dist="uniform", device="cpu")
            if direct:
                m.enable_direct_exchange(ttx_sharded.CollectiveExchange(None))
            if m.local is not None:
                with torch.no_grad():
                    for t, core in enumerate(m.local.tt_cores):
                        core.copy_(torch.from_numpy(cores_all[t][m.my_tables]))
            return m

        a, b = fresh(), fresh()
        batches, grads = [], []
        for k in range(K):
            idx, off, grad = _make_inputs(10 * k + rank, fixed, NT)
            batches.append((torch.from_numpy(idx), torch.from_numpy(off)))
            grads.append(torch.from_numpy(grad))
        outs_a = []
        for (i, o), g in zip(batches, grads):  # every step's index exchange in line
            out = a(i, o, fixed_pooling=fixed)
            outs_a.append(out.detach().clone())
            out.backward(g)
        assert b.prefetch_many(batches, fixed_pooling=fixed) is True and len(b._planned) == K
        same = True
        for k, ((i, o), g) in enumerate(zip(batches, grads)):  # the round's index exchange done up front
            out = b(i, o, fixed_pooling=fixed)
            same = same and torch.equal(out.detach(), outs_a[k])
            out.backward(g)
        assert len(b._planned) == 0
        if a.local is not None:
            for x, y in zip(a.local.tt_cores, b.local.tt_cores):
                same = same and torch.equal(x, y)
        # a batch written to after the planning is an ERROR (a silent in-line exchange on one rank would hang the others);
        # so is a batch that was not planned while planned ones are pending; drop_planned() clears the round
        b.prefetch_many(batches[:2], fixed_pooling=fixed)
        batches[0][0].add_(1).remainder_(int(np.prod(P)))
        for bad in (batches[0], batches[2]):
            try:
                b(*bad, fixed_pooling=fixed)
                same = False
            except RuntimeError as ex:
                same = same and "planned ahead" in str(ex)
        b.drop_planned()
        out_b = b(*batches[0], fixed_pooling=fixed).detach()
        out_a = a(*batches[0], fixed_pooling=fixed).detach()
        same = same and torch.equal(out_a, out_b) and len(b._planned) == 0
        # int32 batches are found by their own tensor objects (the lookup precedes the cast to int64)
        i32 = [(i.int(), o.int()) for i, o in batches[:2]]
        assert b.prefetch_many(i32, fixed_pooling=fixed) is True
        for (i, o) in i32:
            out_b = b(i, o, fixed_pooling=fixed).detach()
            same = same and torch.equal(out_b, a(i, o, fixed_pooling=fixed).detach())
        same = same and len(b._planned) == 0
        assert b.prefetch_many(batches, fixed_pooling=None) is False  # ragged bags: not planned ahead
        q.put((rank, bool(same), "", None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "ERROR", traceback.format_exc(), None))


@pytest.mark.parametrize("NT,direct", [(2, True), (5, True), (5, False), (1, True)])
def test_two_rank_round_planned_ahead_equals_in_line(NT, direct):
    """ShardedTableBatchedTTEmbeddingBag.prefetch_many: ONE index exchange for a round of batches, then the steps with the
    pooled / gradient exchanges only -- outputs and cores identical to exchanging every step's indices in line"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_round, args=(r, world, port, q, NT, direct)) for r in range(world)]
    for p in procs:
        p.start()
    for _ in range(world):
        r = q.get(timeout=180)
        assert r[1] is True, f"rank {r[0]}: {r[1]}\n{r[2]}"
    for p in procs:
        p.join(timeout=60)
