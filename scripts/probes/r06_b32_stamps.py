#!/usr/bin/env python3
"""Shader-clock stamps (ttx_debug_stamps, test build) of bwd32_kernel's work-groups at cfg5's per-GPU shard: where a work-group's
time goes -- staging, the second sub-chunk's prologue / four tiles / epilogue, the fold -- and how the work-groups fill the launch.
    TTX_BWD32=<lookups per chunk> python scripts/probes/r06_b32_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E, tt_embeddings_ops as ops

dev = torch.device("cuda:0")
tables, B = 4, 4096
p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
E_, D = int(np.prod(p)), int(np.prod(q))
m = ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, sparse=True, use_cache=False, weight_dist="uniform", device=dev)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 4, B, tables, 20, E_)]
grad = torch.from_numpy(G.make_grad(2, tables, B, D)).to(dev)
for k in range(4):
    m(*reqs[k]).backward(grad)
nwg = 16384
buf = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
E.debug_stamps(buf.data_ptr())
m(*reqs[1]).backward(grad)
torch.cuda.synchronize()
E.debug_stamps(None)
st = buf.cpu().numpy().reshape(-1, 32).astype(np.float64)
st = st[(st[:, 30] > 0) & (st[:, 0] > 0)]
life = st[:, 30] - st[:, 0]
print(f"{len(st)} (persistent) work-groups; lifetime in clocks of the stamp counter: min {life.min():.0f} med {np.median(life):.0f} max {life.max():.0f}; "
      f"iterations of the sub-chunk loop: min {st[:, 31].min():.0f} med {np.median(st[:, 31]):.0f} max {st[:, 31].max():.0f}")
def row(nm, v):
    v = v[np.isfinite(v)]
    print(f"  {nm:52s} {np.median(v):9.0f} {np.percentile(v, 10):9.0f} {np.percentile(v, 90):9.0f}")
print("wave 0, clocks (median  p10  p90):")
row("entry -> top of the first sub-chunk", st[:, 2] - st[:, 0])
for k in range(7):
    b = 2 + 4 * k
    ok = (st[:, b] > 0) & (st[:, b + 1] > st[:, b]) & (st[:, b + 2] > 0) & (st[:, b + 3] > 0)
    if not ok.any():
        continue
    x = st[ok]
    row(f"sub-chunk {k}: top -> operands arrived and in LDS", x[:, b + 1] - x[:, b])
    row(f"sub-chunk {k}:   -> stream start (stores, phase A of tile 0, requests)", x[:, b + 2] - x[:, b + 1])
    row(f"sub-chunk {k}:   stream", x[:, b + 3] - x[:, b + 2])
    if k < 6:
        ok2 = ok & (st[:, b + 4] > 0)
        row(f"sub-chunk {k}:   stream end -> next top (fold at a chunk end)", st[ok2][:, b + 4] - st[ok2][:, b + 3])
per = life / np.maximum(st[:, 31], 1)
row("lifetime / iterations", per)
