#!/usr/bin/env python3
"""HBM-bound half of the path: achieved GB/s of the cache-row gather (forward), the cache-row SGD
scatter (backward), the hash-table frequency update and the hash lookup + stable partition, against
the 8 TB/s HBM3E peak.  Algorithmic bytes per unit as in SURVEY.md section 8(d).
usage (GPU box): python scripts/bench_cache.py > gpurun_out/cache_bw.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))
import torch
import tt_embeddings as E

dev = torch.device("cuda:0")
D, L, PEAK = 64, 20, 8000.0


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


out = []
# `--only NNZ,ROWS`: one configuration (so that a rocprofv3 --stats average over the run belongs to one size)
only = None
if "--only" in sys.argv:
    only = tuple(int(x) for x in sys.argv[sys.argv.index("--only") + 1].split(","))
for nnz in (10240, 1 << 20):
    for cache_rows in (1 << 18, 1 << 22):
        if only and (nnz, cache_rows) != only:
            continue
        g = torch.Generator(device="cpu").manual_seed(1)
        B = nnz // L
        nnz = B * L  # whole bags
        loc = torch.randint(0, cache_rows, (nnz,), generator=g, dtype=torch.int32).to(dev)
        rowidx = torch.arange(B, dtype=torch.int64).repeat_interleave(L).to(dev)
        w = torch.rand(cache_rows, D, device=dev)
        outp = torch.zeros(B, D, device=dev)
        grad = torch.rand(B, D, device=dev)
        t_f = timed(lambda: E.cache_forward(B, nnz, loc, rowidx, w, outp))
        E.debug_cache_fwd(1)
        t_f_old = timed(lambda: E.cache_forward(B, nnz, loc, rowidx, w, outp))
        E.debug_cache_fwd(0)
        t_b = timed(lambda: E.cache_backward_sgd(nnz, grad, loc, rowidx, 0.0, w, deterministic=False))
        # (round 6) the atomic-free update: stable sort of the lookups by cache row + ordered sums + one writer per row, all launches
        t_bs = timed(lambda: E.cache_backward_sgd(nnz, grad, loc, rowidx, 0.0, w, deterministic=True))
        # ... and on a Zipf(1.2) stream over the rows (what a populated cache sees: row 0 takes a sixth of the lookups)
        import numpy as np
        zl = torch.from_numpy(((np.random.RandomState(3).zipf(1.2, size=nnz) - 1) % cache_rows).astype(np.int32)).to(dev)
        t_bz = timed(lambda: E.cache_backward_sgd(nnz, grad, zl, rowidx, 0.0, w, deterministic=False))
        t_bzs = timed(lambda: E.cache_backward_sgd(nnz, grad, zl, rowidx, 0.0, w, deterministic=True))
        st_ = torch.zeros(cache_rows, device=dev)
        t_az = timed(lambda: E.cache_backward_rowwise_adagrad_approx(nnz, grad, zl, rowidx, 0.0, 1e-4, st_, w, deterministic=False))
        t_azs = timed(lambda: E.cache_backward_rowwise_adagrad_approx(nnz, grad, zl, rowidx, 0.0, 1e-4, st_, w, deterministic=True))
        H = 1 << 22
        idx = torch.randint(0, 11_000_000, (nnz,), generator=g, dtype=torch.int64).to(dev)
        ht = torch.full((H,), -1, dtype=torch.int64, device=dev)
        fr = torch.zeros(H, dtype=torch.int64, device=dev)
        t_u = timed(lambda: E.update_cache_state(idx, ht, fr))
        st = torch.full((H,), -1, dtype=torch.int32, device=dev)
        off = torch.arange(0, nnz + 1, L, dtype=torch.int64, device=dev)
        t_l = timed(lambda: E.preprocess_indices_sync(idx, off, 1, False, ht, st))  # incl. the host read-back
        rec = {"nnz": nnz, "cache_rows": cache_rows, "cache_MiB": cache_rows * D * 4 >> 20}
        for name, t, bytes_ in (("gather_fwd", t_f, nnz * (4 * D + 12) + B * 4 * D),
                                ("gather_fwd_one_group_per_lookup", t_f_old, nnz * (4 * D + 12) + B * 4 * D),
                                ("scatter_sgd_bwd", t_b, nnz * (2 * 4 * D + 12) + B * 4 * D),
                                ("sorted_sgd_bwd_all_launches", t_bs, nnz * (2 * 4 * D + 12) + B * 4 * D),
                                ("scatter_sgd_bwd_zipf", t_bz, nnz * (2 * 4 * D + 12) + B * 4 * D),
                                ("sorted_sgd_bwd_zipf_all_launches", t_bzs, nnz * (2 * 4 * D + 12) + B * 4 * D),
                                ("rowwise_adagrad_bwd_zipf", t_az, nnz * (2 * 4 * D + 12) + B * 4 * D),
                                ("sorted_rowwise_adagrad_bwd_zipf_all_launches", t_azs, nnz * (2 * 4 * D + 12) + B * 4 * D),
                                ("hash_update", t_u, nnz * 24), ("lookup_partition_sync", t_l, nnz * 25 + nnz * 3 * 20)):
            gbs = bytes_ / t / 1e9
            rec[name] = {"us": round(t * 1e6, 2), "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / PEAK, 4)}
        out.append(rec)
print(json.dumps(out, indent=1))
