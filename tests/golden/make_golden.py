#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the REFERENCE's own Python oracle.

Runs only in the build container (needs /root/reference).  It imports the
reference's tt_embeddings_ops.py (with an empty stand-in for the CUDA extension
module it imports at the top; none of its functions is called) and uses exactly
what the reference's tests use as ground truth:

    full = tt_matrix_to_full(p, q, ranks, cores, [1, 0, 2, 3])      (ops.py:80-127)
    out  = F.embedding_bag(indices, full, offsets, mode="sum",
                           include_last_offset=True)               (test.py:95-107)
    core grads by autograd through `full`                          (test.py:161-174)
    SGD     : t - t.grad * lr                                      (test.py:243-246)
    Adagrad : state = g*g ; t - lr*g / (sqrt(state) + eps)         (test.py:317-333)

Inputs come from tests/gen_inputs.py (seeded, numpy legacy stream) and are NOT
stored for the big configs -- the tests regenerate them from the same seeds.
Only data (inputs / expected outputs) is written; no reference source.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen_inputs as G  # noqa: E402

sys.modules.setdefault("tt_embeddings", types.ModuleType("tt_embeddings"))
sys.path.insert(0, "/root/reference")
import tt_embeddings_ops as ref  # noqa: E402

LR, EPS = 0.1, 1.0e-4  # tt_embeddings_test.py:268-269


def reference_case(num_tables, p, q, ranks, cores_np, indices, offsets, d_out):
    T = len(p)
    B = (offsets.size - 1) // num_tables
    cores = [torch.tensor(c, requires_grad=True) for c in cores_np]
    idx = torch.tensor(indices)
    outs = []
    for tb in range(num_tables):
        full = ref.tt_matrix_to_full(p, q, ranks, [c[tb : tb + 1] for c in cores], [1, 0, 2, 3])
        lo, hi = int(offsets[tb * B]), int(offsets[(tb + 1) * B])
        off = torch.tensor(offsets[tb * B : (tb + 1) * B + 1] - lo)
        if hi > lo:
            outs.append(F.embedding_bag(idx[lo:hi], full, off, mode="sum", include_last_offset=True))
        else:
            outs.append(full[:0].sum(0, keepdim=True).expand(B, -1) * 0.0)
    out = torch.stack(outs)  # [tables, B, D]
    out.backward(torch.tensor(d_out))
    grads = [c.grad.detach().clone() for c in cores]
    # optimizer results: exactly the expressions of the reference tests; the small
    # cases store only `grads` (tests re-evaluate these two lines in fp32 numpy,
    # IEEE-identical), the big cases store sub-sampled rows of the results.
    sgd = [(c.detach() - g * LR) for c, g in zip(cores, grads)]
    state = [g * g for g in grads]
    ada = [(c.detach() - torch.div(g * LR, torch.sqrt(s) + EPS)) for c, g, s in zip(cores, grads, state)]
    return out.detach().numpy(), [g.numpy() for g in grads], [s.numpy() for s in sgd], [a.numpy() for a in ada], [s.numpy() for s in state]


def small_cases():
    cases = {}
    cid = 0
    for T in (2, 3, 4):
        p, q, r = G.test_shape(T)
        E = int(np.prod(p))
        for tables in (1, 3):
            for seed in (0, 1):
                dist = "uniform" if seed == 0 else "signed"
                name = f"t{T}_tb{tables}_s{seed}"
                cores = G.make_cores(100 + cid, tables, p, q, r, dist)
                idx, off = G.make_bags(200 + cid, 37, E, 3, 2, tables)
                if idx.size > 4:  # force duplicates inside a bag and across bags
                    idx[1] = idx[0]
                    idx[-1] = idx[0]
                d_out = G.make_grad(300 + cid, tables, 37, int(np.prod(q)))
                cases[name] = (tables, p, q, r, cores, idx, off, d_out)
                cid += 1
    # README toy example (BASELINE config 1): E=10, D=3, ranks [2,2]
    c = G.CFG1
    cores = G.make_cores(7, 1, c["p"], c["q"], c["ranks"], "signed")
    idx = np.array([1, 2, 4, 5, 4, 3, 2, 9], dtype=np.int64)
    off = np.array([0, 4, 8], dtype=np.int64)
    cases["cfg1_toy"] = (1, c["p"], c["q"], c["ranks"], cores, idx, off, G.make_grad(8, 1, 2, 3))
    return cases


def round4_cases():
    """the geometry classes round 4 moved onto new routes, as small tables the reference expands: a first factor with no exact part
    split (q0 = 5, 7: core 0 zero-padded by the module), q0 = 8 (part lookups), q2 = 12 / 16 (the q2 <= 16 templates, padded and
    exact, ranks 32 and 64), four cores with a merged last factor of 8 and of 16, two cores at a rank that is not a multiple of 4"""
    shapes = {
        "t3_q5": ([6, 7, 8], [5, 4, 4], [16, 16]),
        "t3_q7": ([5, 6, 7], [7, 4, 8], [16, 32]),
        "t3_q8": ([6, 7, 8], [8, 4, 4], [16, 16]),
        "t3_q2_12": ([5, 6, 7], [4, 4, 12], [32, 32]),
        "t3_q2_16_r64": ([3, 3, 3], [4, 8, 16], [64, 64]),
        "t4_m8": ([4, 5, 3, 4], [2, 4, 4, 2], [16, 16, 16]),
        "t4_m16": ([4, 5, 3, 4], [2, 4, 4, 4], [16, 16, 16]),
        "t2_r13": ([9, 8], [5, 7], [13]),
        "t2_r6": ([9, 8], [16, 3], [6]),
    }
    cases = {}
    for cid, (name, (p, q, r)) in enumerate(shapes.items()):
        for tables in ((1,) if max(r) > 32 else (1, 2)):  # (the r = 64 cores are a megabyte per table)
            E = int(np.prod(p))
            cores = G.make_cores(400 + 2 * cid + tables, tables, p, q, r, "signed")
            idx, off = G.make_bags(500 + 2 * cid + tables, 29, E, 3, 2, tables)
            if idx.size > 4:
                idx[1] = idx[0]
                idx[-1] = idx[0]
            d_out = G.make_grad(600 + 2 * cid + tables, tables, 29, int(np.prod(q)))
            cases[f"{name}_tb{tables}"] = (tables, p, q, r, cores, idx, off, d_out)
    return cases


def write_small(cases=None, fname="small_cases.npz"):
    blob = {}
    for name, (tables, p, q, r, cores, idx, off, d_out) in (cases or small_cases()).items():
        out, grads, sgd, ada, state = reference_case(tables, p, q, r, cores, idx, off, d_out)
        blob[f"{name}/meta"] = np.array([tables, len(p)] + list(p) + list(q) + list(r), dtype=np.int64)
        blob[f"{name}/indices"] = idx
        blob[f"{name}/offsets"] = off
        blob[f"{name}/d_out"] = d_out
        blob[f"{name}/out"] = out
        for t in range(len(p)):
            blob[f"{name}/core{t}"] = cores[t]
            blob[f"{name}/grad{t}"] = grads[t]
        print(name, "nnz", idx.size, "out", out.shape)
    np.savez_compressed(os.path.join(HERE, fname), **blob)


def write_big(tag, cfg, seed, nrows=16):
    p, q, r, B, L = cfg["p"], cfg["q"], cfg["ranks"], cfg["B"], cfg["L"]
    E, D = int(np.prod(p)), int(np.prod(q))
    cores = G.make_cores(seed, 1, p, q, r, "uniform")
    idx, off = G.make_requests(seed + 1, 1, B, 1, L, E)[0]
    d_out = G.make_grad(seed + 2, 1, B, D)
    out, grads, sgd, ada, state = reference_case(1, p, q, r, cores, idx, off, d_out)
    rs = np.random.RandomState(seed + 3)
    blob = {"seed": np.array([seed]), "out": out}
    for t in range(3):
        g = grads[t].reshape(-1, grads[t].shape[-1])  # [p_t, slice]
        rows = np.sort(rs.choice(g.shape[0], size=min(nrows, g.shape[0]), replace=False))
        blob[f"grad{t}_rows"] = rows
        blob[f"grad{t}_sub"] = g[rows]
        blob[f"sgd{t}_sub"] = sgd[t].reshape(g.shape)[rows]
        blob[f"ada{t}_sub"] = ada[t].reshape(g.shape)[rows]
        blob[f"grad{t}_sum"] = np.array([g.astype(np.float64).sum(), (g.astype(np.float64) ** 2).sum()])
        blob[f"grad{t}_rowsum"] = g.astype(np.float64).sum(axis=1)  # per-slice sums (p_t doubles)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **blob)
    print(tag, "out", out.shape, [float(blob[f"grad{t}_sum"][0]) for t in range(3)])


def write_cfg5():
    """BASELINE configs[4] (26 tables of cfg2's shape) at B = 512: the reference's results for three of the 26
    tables (tables are independent; each is a full 11M x 64 expansion)"""
    c = G.cfg5_case(512)
    blob = {"seed": np.array([G.CFG5_SEED]), "tables": np.array(G.CFG5_GOLDEN_TABLES)}
    rs = np.random.RandomState(G.CFG5_SEED + 3)
    for k in G.CFG5_GOLDEN_TABLES:
        ck = G.table_of(c, k)
        out, grads, sgd, ada, state = reference_case(1, ck["p"], ck["q"], ck["r"], [np.ascontiguousarray(x) for x in ck["cores"]],
                                                     ck["indices"], ck["offsets"], ck["d_out"])
        blob[f"t{k}_out"] = out[0]
        for t in range(3):
            g = grads[t].reshape(-1, grads[t].shape[-1])
            rows = np.sort(rs.choice(g.shape[0], size=8, replace=False))
            blob[f"t{k}_grad{t}_rows"] = rows
            blob[f"t{k}_grad{t}_sub"] = g[rows]
            blob[f"t{k}_sgd{t}_sub"] = sgd[t].reshape(g.shape)[rows]
            blob[f"t{k}_grad{t}_rowsum"] = g.astype(np.float64).sum(axis=1)
        print("cfg5 table", k, "out", out.shape)
    np.savez_compressed(os.path.join(HERE, "cfg5.npz"), **blob)


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--round4" in sys.argv:  # (only the round-4 file: the others are not touched)
        write_small(round4_cases(), "round4_cases.npz")
        sys.exit(0)
    write_small()
    write_small(round4_cases(), "round4_cases.npz")
    if "--small" not in sys.argv:
        write_big("cfg2", G.CFG2, 1234)
        write_big("cfg4", G.CFG4, 4321)
        write_cfg5()
        write_big("r128", G.R128, 2468, nrows=2)  # (a core_1 slice is 256 KB at r = 128: two sampled slices per core)
