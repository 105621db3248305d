#!/usr/bin/env python3
"""Per-phase wall-clock stamps of the backward contraction work-groups at cfg2
(debug facility ttx_debug_stamps).  Prints, relative to the first work-group's
entry, the distribution of each phase boundary and of each phase's duration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E, tt_embeddings_ops as ops

dev = torch.device("cuda:0")
p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
E_, D = int(np.prod(p)), int(np.prod(q))
m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, use_cache=False, weight_dist="uniform", device=dev)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 10, 512, 1, 20, E_)]
grad = torch.from_numpy(G.make_grad(2, 1, 512, D)[0]).to(dev)
for k in range(5):
    m(*reqs[k]).backward(grad)
buf = torch.zeros(1100 * 16, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
E.debug_stamps(buf.data_ptr())
m(*reqs[5]).backward(grad)
torch.cuda.synchronize()
E.debug_stamps(None)
st = buf.cpu().numpy().reshape(-1, 16)[:1000]
live = st[:, 9] > 0
st = st[live][:, :10].astype(np.float64) / 100.0  # 100 MHz -> us
t0 = st[:, 0].min()
names = ["entry", "chunk_rec", "B1 staged+sync", "group recs", "A staged", "G/c2 issued", "x0 in regs", "tail+dX0 stored", "groups done", "dB1 reduced (end)"]
print(f"{live.sum()} work-groups (wave 0's first group); us relative to the first entry")
for i, nm in enumerate(names):
    v = st[:, i] - t0
    print(f"  {nm:20s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f}")
raw = buf.cpu().numpy().reshape(-1, 16)[:1000][live].astype(np.float64) / 100.0
print(f"probe: lrec vector load latency  min {(raw[:,11]-raw[:,10]).min():.2f} med {np.median(raw[:,11]-raw[:,10]):.2f} max {(raw[:,11]-raw[:,10]).max():.2f} us")
print(f"probe: B1   vector load latency  min {(raw[:,12]-raw[:,11]).min():.2f} med {np.median(raw[:,12]-raw[:,11]):.2f} max {(raw[:,12]-raw[:,11]).max():.2f} us")
print("phase durations (per work-group):")
for i in range(1, 10):
    v = st[:, i] - st[:, i - 1]
    print(f"  {names[i-1]:>20s} -> {names[i]:20s} min {v.min():6.2f}  med {np.median(v):6.2f}  max {v.max():6.2f}")
pl = buf.cpu().numpy().reshape(-1, 16)[1000:1003, :9].astype(np.float64) / 100.0
pl = pl[:, [0, 8, 1, 6, 7, 2, 3, 4, 5]]
pn = ["entry", "loads back", "decoded", "counted", "scanned", "sorted", "perm/off stored", "lrec stored", "chunk list"]
print("plan kernel (one work-group per core), us since that block's entry:")
for b in range(3):
    print(f"  core {b}: " + "  ".join(f"{pn[i]}={pl[b, i] - pl[b, 0]:.2f}" for i in range(1, 9) if pl[b, i] > 0))
