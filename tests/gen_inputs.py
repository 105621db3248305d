"""Seeded synthetic inputs shared by the golden-vector script, the parity tests
and bench.py.  Everything is generated on the CPU with numpy's frozen legacy
RandomState stream so the build container (where the golden vectors are made
with the reference's Python oracle) and the GPU box see identical data.

Distributions follow the reference's own harness:
  * bags: tt_embeddings_test.py:22-50 (generate_sparse_feature: lengths
    round(N(pf, std)) clipped at 0, indices uniform with replacement);
  * requests: tt_embeddings_benchmark.py:37-91 (uniform randint, or zipf % E);
  * cores: reset_parameters "uniform" / "normal", tt_embeddings_ops.py:621-641.
"""
import numpy as np


def pad_ranks(ranks, T):
    ranks = [int(x) for x in ranks]
    return ranks if len(ranks) == T + 1 else [1] + ranks + [1]


def core_shapes(num_tables, p, q, ranks):
    T = len(p)
    r = pad_ranks(ranks, T)
    return [(num_tables, int(p[t]), r[t] * int(q[t]) * r[t + 1]) for t in range(T)]


def make_cores(seed, num_tables, p, q, ranks, dist="uniform"):
    T = len(p)
    r = pad_ranks(ranks, T)
    E = int(np.prod(np.array(p, dtype=np.int64)))
    D = int(np.prod(np.array(q, dtype=np.int64)))
    rs = np.random.RandomState(seed)
    cores = []
    for shp in core_shapes(num_tables, p, q, r):
        if dist == "uniform":
            stddev = np.sqrt(2.0 / (E + D))
            var = np.prod(np.array(r, dtype=np.float64) ** (-1.0 / (2 * T)))
            hi = stddev ** (1.0 / T) * var
            c = rs.uniform(0.0, hi, size=shp)
        elif dist == "normal":
            c = rs.normal(0.0, 1.0 / np.sqrt(E), size=shp) * (1.0 / r[0])
        elif dist == "signed":  # well-scaled, sign-mixed (exercises cancellation)
            c = rs.uniform(-1.0, 1.0, size=shp) / np.sqrt(max(shp[2] // max(q[len(cores)], 1), 1))
        else:
            raise ValueError(dist)
        cores.append(np.ascontiguousarray(c, dtype=np.float32))
    return cores


def make_bags(seed, batch_size, num_embeddings, pooling_factor, pooling_factor_std, num_tables=1):
    """-> (indices int64[nnz], offsets int64[num_tables*batch_size+1]); bags are
    table-major like TableBatchedTTEmbeddingBag.forward expects."""
    rs = np.random.RandomState(seed)
    lengths = np.round(rs.normal(pooling_factor, pooling_factor_std, batch_size * num_tables)).astype(np.int64)
    lengths = np.where(lengths < 0, 0, lengths)
    total = int(lengths.sum())
    indices = rs.randint(0, num_embeddings, size=total).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    return indices, offsets


def make_requests(seed, iters, B, num_tables, L, E, alpha=1.0):
    """benchmark request stream: iters x (indices[num_tables*B*L], offsets)."""
    rs = np.random.RandomState(seed)
    out = []
    offsets = np.arange(0, num_tables * B * L + 1, L, dtype=np.int64)
    for _ in range(iters):
        if alpha <= 1.0:
            idx = rs.randint(0, E, size=num_tables * B * L).astype(np.int64)
        else:
            idx = (rs.zipf(a=alpha, size=num_tables * B * L).astype(np.int64)) % E
        out.append((idx, offsets.copy()))
    return out


def make_grad(seed, num_tables, B, D):
    rs = np.random.RandomState(seed)
    return (rs.rand(num_tables, B, D) * 0.1).astype(np.float32)  # tt_embeddings_test.py:160


# ---- named cases -----------------------------------------------------------
TEST_P = [7, 9, 11, 5]      # tt_embeddings_test.py:65-70
TEST_Q = [3, 4, 5, 7]
TEST_R = [13, 12, 7]


def test_shape(T):
    return TEST_P[:T], TEST_Q[:T], TEST_R[: T - 1]


CFG1 = dict(p=[1, 2, 5], q=[1, 1, 3], ranks=[2, 2])                  # README toy example
CFG2 = dict(p=[200, 220, 250], q=[4, 4, 4], ranks=[32, 32], B=512, L=20)  # benchmark default
CFG4 = dict(p=[200, 220, 250], q=[4, 4, 8], ranks=[64, 64], B=512, L=20)
# ranks 128 (the r = 128 shape-specialised kernels; not a BASELINE config): a table small enough to expand with the reference
R128 = dict(p=[20, 22, 25], q=[4, 4, 4], ranks=[128, 128], B=96, L=8)
CFG5 = dict(p=[200, 220, 250], q=[4, 4, 4], ranks=[32, 32], tables=26, L=20)  # 26-table batched lookup, global B = 4096
CFG5_SEED = 5150
CFG5_GOLDEN_TABLES = (0, 13, 25)  # tables whose reference results are stored in tests/golden/cfg5.npz (B = 512)


def cfg5_case(B, seed=CFG5_SEED):
    """BASELINE configs[4] on one device: 26 tables of cfg2's shape, B bags per table, 20 lookups per bag"""
    c = CFG5
    p, q, r = c["p"], c["q"], pad_ranks(c["ranks"], 3)
    E, D = int(np.prod(p)), int(np.prod(q))
    idx, off = make_requests(seed + 1 + B, 1, B, c["tables"], c["L"], E)[0]
    return dict(tables=c["tables"], T=3, p=p, q=q, r=r, B=B, D=D, indices=idx, offsets=off,
                cores=make_cores(seed, c["tables"], p, q, r, "uniform"), d_out=make_grad(seed + 2, c["tables"], B, D))


def table_of(c, k):
    """the one-table case made of table k of a table-batched case"""
    B = c["B"]
    lo, hi = int(c["offsets"][k * B]), int(c["offsets"][(k + 1) * B])
    return dict(tables=1, T=c["T"], p=c["p"], q=c["q"], r=c["r"], B=B, D=c["D"], indices=c["indices"][lo:hi],
                offsets=c["offsets"][k * B:(k + 1) * B + 1] - lo, cores=[x[k:k + 1] for x in c["cores"]],
                d_out=c["d_out"][k:k + 1])
