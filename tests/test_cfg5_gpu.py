"""BASELINE configs[4] on one device: the 26-table batched lookup (26 tables of cfg2's shape, p = [200, 220, 250],
q = [4, 4, 4], ranks 32).  6500 slice ids per core -> the table-group plan of the module's route
(ttx_lookup_prologue), more than 131,072 lookups -> the sub-chunk (MULTI) variant of the specialised kernels.

* B = 512 per table (266,240 lookups): output, dense core gradients and fused SGD against the CPU oracle IN FULL
  and against the reference's own results for three of the tables (tests/golden/cfg5.npz).
* B = 4096 per table (the global batch of configs[4], 2.13 M lookups): whole tables against the oracle (tables are
  independent, so table k of the batched run must equal a one-table oracle run), bit-identical from run to run,
  gradient exactly linear in d_output."""
import numpy as np
import pytest
import torch

import gen_inputs as G
import oracle_lib as O
from test_oracle_golden import check_cfg5_tables
from util import LR, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def module_for(c, **kw):
    import tt_embeddings_ops as ops

    m = ops.TableBatchedTTEmbeddingBag(c["tables"], int(np.prod(c["p"])), c["D"], c["r"][1:-1], c["p"], c["q"],
                                       weight_dist="uniform", use_cache=False, device=DEV, **kw)
    with torch.no_grad():
        for dst, src in zip(m.tt_cores, c["cores"]):
            dst.copy_(t(src))
    return m


def oracle_table(c, k, mode):
    ck = G.table_of(c, k)
    cores = [np.ascontiguousarray(x).copy() for x in ck["cores"]]
    g = O.make_geom(1, c["p"], c["q"], c["r"])
    rowidx, tableidx = O.rowidx_from_offsets(ck["offsets"], 1)
    out = O.tt_forward(g, c["B"], c["D"], ck["indices"], rowidx, tableidx, cores)[0]
    if mode == "dense":
        return out, [x[0] for x in O.tt_backward(g, O.OPTIM_DENSE, c["B"], c["D"], 0, 0, ck["indices"], rowidx, tableidx, ck["d_out"], cores)]
    O.tt_backward(g, O.OPTIM_SGD, c["B"], c["D"], LR, 0, ck["indices"], rowidx, tableidx, ck["d_out"], cores)
    return out, [x[0] for x in cores]


def test_cfg5_b512_in_full_vs_oracle_and_reference_golden():
    import tt_embeddings_ops as ops

    c = G.cfg5_case(512)
    assert c["indices"].size == 26 * 512 * 20
    m = module_for(c, sparse=False)
    out = m(t(c["indices"]), t(c["offsets"]))
    out.backward(t(c["d_out"]))
    got_out = out.detach().cpu().numpy()
    got_g = [x.grad.cpu().numpy() for x in m.tt_cores]
    ms = module_for(c, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)
    ms(t(c["indices"]), t(c["offsets"])).backward(t(c["d_out"]))
    got_sgd = [x.detach().cpu().numpy() for x in ms.tt_cores]
    # the whole batched case in ONE oracle call (table-major bags, tableidx from the offsets)
    g = O.make_geom(26, c["p"], c["q"], c["r"])
    rowidx, tableidx = O.rowidx_from_offsets(c["offsets"], 26)
    ref_out = O.tt_forward(g, 512, 64, c["indices"], rowidx, tableidx, c["cores"])
    ref_g = O.tt_backward(g, O.OPTIM_DENSE, 512, 64, 0, 0, c["indices"], rowidx, tableidx, c["d_out"], [x.copy() for x in c["cores"]])
    assert_close(got_out, ref_out, "cfg5 B=512 out")
    for k in range(3):
        assert_close(got_g[k], ref_g[k], f"cfg5 B=512 grad{k}")
        assert_close(got_sgd[k], c["cores"][k] - np.float32(LR) * ref_g[k], f"cfg5 B=512 sgd core{k}")
    check_cfg5_tables(c, got_out, got_g, got_sgd)


def test_cfg5_global_batch_4096_tables_vs_oracle_determinism_linearity():
    import tt_embeddings_ops as ops

    c = G.cfg5_case(4096)
    assert c["indices"].size == 26 * 4096 * 20
    m = module_for(c, sparse=False)
    idx, off, d_out = t(c["indices"]), t(c["offsets"]), t(c["d_out"])

    def run(scale):
        for x in m.tt_cores:
            x.grad = None
        out = m(idx, off)
        out.backward(d_out * scale)
        return out.detach().clone(), [x.grad.clone() for x in m.tt_cores]

    out1, g1 = run(1.0)
    out1b, g1b = run(1.0)
    _, g2 = run(2.0)
    assert torch.equal(out1, out1b), "forward is not run-to-run deterministic"
    for a, b, d in zip(g1, g1b, g2):
        assert torch.equal(a, b), "backward is not run-to-run deterministic"
        assert torch.equal(a * 2.0, d), "gradient is not exactly linear in d_output"
    ms = module_for(c, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=LR)
    ms(idx, off).backward(d_out)
    for k in (3, 25):
        ref_out, ref_g = oracle_table(c, k, "dense")
        assert_close(out1[k].cpu().numpy(), ref_out, f"cfg5 B=4096 table {k} out")
        _, ref_sgd = oracle_table(c, k, "sgd")
        for cidx in range(3):
            assert_close(g1[cidx][k].cpu().numpy(), ref_g[cidx], f"cfg5 B=4096 table {k} grad{cidx}")
            assert_close(ms.tt_cores[cidx][k].detach().cpu().numpy(), ref_sgd[cidx], f"cfg5 B=4096 table {k} sgd core{cidx}")
    # every other table: bag sums of a sub-sample of bags through the oracle's row decompression
    rs = np.random.RandomState(7)
    g1t = O.make_geom(1, c["p"], c["q"], c["r"])
    for k in range(26):
        bags = rs.choice(4096, size=24, replace=False)
        cores_k = [np.ascontiguousarray(x[k:k + 1]) for x in c["cores"]]
        for b in bags:
            lo, hi = int(c["offsets"][k * 4096 + b]), int(c["offsets"][k * 4096 + b + 1])
            rows = O.tt_rows(g1t, 64, c["indices"][lo:hi], None, cores_k)
            ref = np.zeros(64, dtype=np.float32)
            for rrow in rows:  # index order, like reduce_output_kernel (cu:920-962)
                ref += rrow
            assert_close(out1[k, b].cpu().numpy(), ref, f"cfg5 B=4096 table {k} bag {b}")
