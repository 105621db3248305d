#!/usr/bin/env python3
"""Phase stamps (ttx_debug_stamps) of the backward contraction work-groups at a LARGE batch (default cfg5's per-GPU shard):
the timeline of the LAST sub-chunk of every work-group, and when the work-groups start and end relative to the launch.
    python scripts/phase_times_large.py [tables] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E, tt_embeddings_ops as ops

dev = torch.device("cuda:0")
tables = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
E_, D = int(np.prod(p)), int(np.prod(q))
kw = dict(sparse=True, use_cache=False, weight_dist="uniform", device=dev)
m = ops.TTEmbeddingBag(E_, D, r, p, q, **kw) if tables == 1 else ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, **kw)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 4, B, tables, 20, E_)]
g = G.make_grad(2, tables, B, D)
grad = torch.from_numpy(g[0] if tables == 1 else g).to(dev)
for k in range(4):
    m(*reqs[k]).backward(grad)
nwg = 16384
buf = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
E.debug_stamps(buf.data_ptr())
m(*reqs[1]).backward(grad)
torch.cuda.synchronize()
E.debug_stamps(None)
st = buf.cpu().numpy().reshape(-1, 16)
live = (st[:, 9] > 0) & (st[:, 0] > 0)
st = st[live].astype(np.float64) / 100.0  # 100 MHz -> us
t0 = st[:, 0].min()
print(f"{live.sum()} work-groups; kernel span {st[:, 9].max() - t0:.1f} us")
dur = st[:, 9] - st[:, 0]
print(f"work-group lifetime: min {dur.min():.2f} med {np.median(dur):.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f} us")
mid = (st[:, 0] - t0 > 0.25 * (st[:, 9].max() - t0)) & (st[:, 0] - t0 < 0.6 * (st[:, 9].max() - t0)) & (st[:, 10] > 0)
s2 = st[mid]
print(f"{mid.sum()} work-groups that start in the middle of the kernel; their LAST sub-chunk (wave 0, group 0), us:")
seq = [(10, "sub-chunk top"), (3, "loads issued"), (2, "B1 staged + sync"), (4, "A/G landed + in LDS"), (7, "GEMM1 + tail + dX0 stored"),
       (5, "d core_0 GEMM (pass 0)"), (6, "coop d core_1 (pass 0) + sync"), (8, "pass 1 up to d core_0"), (9, "coop d core_1 (pass 1) + stores")]
prev = None
for idx, nm in seq:
    if prev is not None:
        v = s2[:, idx] - s2[:, prev]
        print(f"  -> {nm:34s} med {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}")
    prev = idx
v = s2[:, 9] - s2[:, 10]
print(f"  whole last sub-chunk                  med {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}")
full = mid & (st[:, 14] > 0)
s3 = st[full]
if len(s3):
    print(f"{full.sum()} of them with four sub-chunks: entry -> top of sub-chunk 0 / 1 / 2 / 3 -> end, us (median)")
    pts = [s3[:, 0], s3[:, 11], s3[:, 12], s3[:, 13], s3[:, 14], s3[:, 9]]
    print("   " + "  ".join(f"{np.median(pts[i + 1] - pts[i]):6.2f}" for i in range(5)))
# concurrency: how many work-groups are alive at a time
ev = np.concatenate([np.stack([st[:, 0], np.ones(len(st))], 1), np.stack([st[:, 9], -np.ones(len(st))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
print(f"work-groups alive: max {alive.max():.0f}, time-weighted mean {np.sum(alive[:-1] * np.diff(ev[:, 0])) / (ev[-1, 0] - ev[0, 0]):.0f} (256 CUs)")
