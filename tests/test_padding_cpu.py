"""CPU tests (no GPU) of the round-4 zero-padding logic above the C ABI: which first factors q0 are padded to what
(`_pad0_target`), the padded copy of core 0 the module keeps beside its Parameter (`_padded0` / `_padded0_store`), and tables of
different factorings q in one launch set (`VarTableTTEmbeddingBag(table_q=)`, `MixedTTEmbeddingBag(pad_q=)`): grouping, the
per-table column map, and -- by a plain numpy contraction of the PADDED cores -- that a table's own values come out of its
padded rows and nothing else does.  The kernels themselves run these geometries under `-m gpu` (tests/test_module_gpu.py)."""
import types

import numpy as np
import pytest
import torch

import tt_embeddings_ops as ops
import ttx_mixed


def test_which_first_factors_are_padded():
    r = [32, 32]
    got = {q0: (ops._split0_factor([q0, 8, 8], r), ops._pad0_target([q0, 8, 8], r)) for q0 in range(2, 18)}
    # q0 <= 4: the templates hold it; 6, 8, 9, 12, 16: exact part splits (k parts of 2..4 slots); the rest: padded to a multiple of 4
    assert got == {2: (0, 0), 3: (0, 0), 4: (0, 0), 5: (0, 8), 6: (2, 0), 7: (0, 8), 8: (2, 0), 9: (3, 0), 10: (0, 12), 11: (0, 12),
                   12: (3, 0), 13: (0, 16), 14: (0, 16), 15: (0, 16), 16: (4, 0), 17: (0, 0)}
    # q2 beyond 8 has templates at ranks <= 64 only; T != 3 never splits
    assert ops._pad0_target([5, 8, 16], [64, 64]) == 8 and ops._pad0_target([5, 8, 16], [128, 128]) == 0
    assert ops._pad0_target([5, 8], [32]) == 0 and ops._split0_factor([8, 8, 8, 8], [8, 8, 8]) == 0


def test_padded_copy_of_core_0_follows_the_parameter():
    """`_padded0`: refreshed when the Parameter / the state buffer was written in place or re-bound, left alone otherwise;
    `_padded0_store`: the real slots go back.  (The methods only touch the attributes set up here.)"""
    T = ops.TableBatchedTTEmbeddingBag
    ns = types.SimpleNamespace()
    ns.tt_cores = [torch.nn.Parameter(torch.randn(2, 3, 5 * 4))]
    ns.optimizer_state = [torch.rand(2, 3, 20)]
    ns.tt_q_shapes, ns.tt_ranks = [5, 8, 8], [1, 4, 4, 1]
    c, s = T._padded0(ns, 8, 1, True)
    assert c.shape == (2, 3, 32) and s.shape == (2, 3, 32) and c.requires_grad
    assert torch.equal(c[:, :, :20], ns.tt_cores[0]) and (c[:, :, 20:] == 0).all() and torch.equal(s[:, :, :20], ns.optimizer_state[0])
    with torch.no_grad():  # what a fused optimizer does to the padded copies
        c[:, :, :20] += 1
        s[:, :, :20] += 2
    w0, s0 = ns.tt_cores[0].detach().clone(), ns.optimizer_state[0].clone()
    T._padded0_store(ns)
    assert torch.equal(ns.tt_cores[0], w0 + 1) and torch.equal(ns.optimizer_state[0], s0 + 2)
    c2, _ = T._padded0(ns, 8, 1, True)
    assert c2 is c, "nothing was written since the write-back: no refresh"
    with torch.no_grad():
        ns.tt_cores[0].mul_(2)  # an in-place write bumps the version
    c3, _ = T._padded0(ns, 8, 1, True)
    assert torch.equal(c3[:, :, :20], ns.tt_cores[0]) and (c3[:, :, 20:] == 0).all()
    ns.tt_cores[0] = torch.nn.Parameter(torch.randn(2, 3, 20))  # re-bound
    c4, _ = T._padded0(ns, 8, 1, True)
    assert torch.equal(c4[:, :, :20], ns.tt_cores[0])
    d, none = T._padded0(ns, 8, 2, False)  # dense gradients: an autograd pad of the Parameter
    assert none is None and d.requires_grad and d.shape == (2, 3, 32)
    d.sum().backward()
    assert torch.equal(ns.tt_cores[0].grad, torch.ones_like(ns.tt_cores[0]))


def _row(cores, q, r, idx):
    x = cores[0][idx[0]].reshape(q[0], r[1])
    x = (x @ cores[1][idx[1]].reshape(r[1], q[1] * r[2])).reshape(q[0] * q[1], r[2])
    return (x @ cores[2][idx[2]].reshape(r[2], q[2])).reshape(-1)


@pytest.mark.parametrize("pad_q,groups", [(True, [[0, 1, 2]]), (None, [[0, 1], [2]]), (False, [[0, 1], [2]])])
def test_tables_of_different_factorings_in_one_group(pad_q, groups):
    Es, ps = [9000, 8000, 64000], [[20, 22, 25], [20, 22, 25], [40, 40, 40]]
    ranks, qs = [[32, 32], [16, 16], [16, 16]], [[4, 4, 4], [4, 4, 4], [2, 4, 8]]
    m = ttx_mixed.MixedTTEmbeddingBag(Es, 64, ranks, ps, qs, fused=True, pad_q=pad_q, device=torch.device("cpu"), weight_dist="uniform")
    # (auto: ranks 32 / 16 / 16 padded to 32 AND q padded to [4,4,8] would cost more than twice the tables' own work)
    assert m.group_tables == groups
    if not pad_q:
        return
    g = m.groups[0]
    assert g.tt_q_shapes == [4, 4, 8] and g.embedding_dim == 128 and g.out_dim == 64 and g.tt_ranks == [1, 32, 32, 1]
    for k, q in enumerate(qs):
        nat = torch.arange(64).reshape(q)
        pad = torch.full(g.tt_q_shapes, -1)
        pad[:q[0], :q[1], :q[2]] = nat
        assert torch.equal(pad.reshape(-1)[g._cols[k]], torch.arange(64)), f"column map of table {k}"
        rk = [1] + ranks[k] + [1]
        own = _row([g.table_core(k, t) for t in range(3)], q, rk, (1, 2, 3))
        padded = _row([g.table_rows(t)[k] for t in range(3)], g.tt_q_shapes, g.tt_ranks, (1, 2, 3))
        assert torch.allclose(padded[g._cols[k]], own, rtol=1e-6, atol=1e-9), f"table {k}: its values out of the padded row"
        rest = torch.ones(128, dtype=torch.bool)
        rest[g._cols[k]] = False
        assert (padded[rest] == 0).all(), f"table {k}: the padded positions are zero"
        for t in range(3):  # natural shape in, natural shape out
            c = g.table_core(k, t).clone()
            g.set_table_core(k, t, c * 2)
            assert torch.equal(g.table_core(k, t), c * 2) and int((g.table_rows(t)[k] != 0).sum()) == int((c != 0).sum())


def test_equal_ranks_different_factorings_group_by_default():
    Es, ps = [9000, 8000, 64000], [[20, 22, 25], [20, 22, 25], [40, 40, 40]]
    m = ttx_mixed.MixedTTEmbeddingBag(Es, 64, [[32, 32]] * 3, ps, [[4, 4, 4], [4, 4, 4], [2, 4, 8]], fused=True,
                                      device=torch.device("cpu"), weight_dist="uniform")
    assert m.group_tables == [[0, 1, 2]] and m.groups[0].table_q == [[4, 4, 4], [4, 4, 4], [2, 4, 8]] and m.groups[0].table_ranks is None


def test_padding_is_decided_per_group_of_equal_core_count():
    """Round 4 advisor: the auto rule was ONE flag -- three-core tables that pass the 2x test switched q- and rank-padding on for
    the two-core tables as well (here: [8,8] r = 32 and [2,32] r = 4 would have become [8,32] r = 32: 8x the multiply-adds, and a
    factoring outside the specialised templates).  Now every group of equal core count decides for itself."""
    Es = [9000, 8000, 64000, 5000, 6000]
    ps = [[20, 22, 25], [20, 22, 25], [40, 40, 40], [70, 80], [80, 90]]
    ranks = [[32, 32], [32, 32], [32, 32], [32], [4]]
    qs = [[4, 4, 4], [4, 4, 4], [2, 4, 8], [8, 8], [2, 32]]
    m = ttx_mixed.MixedTTEmbeddingBag(Es, 64, ranks, ps, qs, fused=True, device=torch.device("cpu"), weight_dist="uniform")
    assert m.group_tables == [[0, 1, 2], [3], [4]]
    assert m.groups[0].table_q == [[4, 4, 4], [4, 4, 4], [2, 4, 8]]
    assert m.groups[1].tt_q_shapes == [8, 8] and m.groups[2].tt_q_shapes == [2, 32] and m.groups[2].tt_ranks == [1, 4, 1]
    # asked for explicitly, every group pads
    m2 = ttx_mixed.MixedTTEmbeddingBag(Es, 64, ranks, ps, qs, fused=True, pad_q=True, device=torch.device("cpu"), weight_dist="uniform")
    assert m2.group_tables == [[0, 1, 2], [3, 4]] and m2.groups[1].tt_q_shapes == [8, 32] and m2.groups[1].tt_ranks == [1, 32, 1]
