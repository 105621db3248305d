"""diagnostic: cache-live route, free-running vs synchronised: where do two modules of one state part ways?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tt_embeddings_ops as ops, gen_inputs as G, tt_embeddings as E
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
p, q, r = [20, 22, 25], [4, 4, 4], [16, 16]
E_, D, B = 20 * 22 * 25, 64, 64
batches = []
for step in range(4):
    idx, off = G.make_bags(500 + step, B, E_, 6, 3, 1)
    batches.append((t(idx), t(off), t(G.make_grad(600 + step, 1, B, D)[0])))
def fresh():
    m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, eps=1e-3, weight_dist="uniform",
                           device=DEV, use_cache=True, cache_size=256, hashtbl_size=4096)
    with torch.no_grad():
        for dst, src in zip(m.tt_cores, G.make_cores(95, 1, p, q, [1] + r + [1], "signed")):
            dst.copy_(t(src))
    return m
ops._DIRECT_BACKWARD = bool(int(os.environ.get("DIRECT", "0")))
SYNC_POP = bool(int(os.environ.get("SYNC_POP", "0")))
SYNC_MASK = int(os.environ.get("SYNC_MASK", "4"))
a = fresh()
with torch.no_grad():
    for i, o, _ in batches: a(i, o)
a.cache_populate()
if SYNC_POP: torch.cuda.synchronize()
mods = [a]
for _ in range(2):
    b = fresh(); b.load_state_dict(a.state_dict()); b.warmup = False; mods.append(b)
if SYNC_POP: torch.cuda.synchronize()
hist = []
for k, m in enumerate(mods):
    h = []
    for i, o, g in batches:
        out = m(i, o)
        h.append([out.detach().clone()])
        out.backward(g)
        h[-1] += [c.detach().clone() for c in m.tt_cores] + [m.cache_weight.detach().clone()]
        if (SYNC_MASK >> k) & 1: torch.cuda.synchronize()
    hist.append(h)
torch.cuda.synchronize()
names = ["out", "core0", "core1", "core2", "cache_rows"]
for k in (0, 1):
    for s in range(4):
        print(f"module {k} vs synchronised, step {s}:", {n: f"{float((x - y).abs().max()):.2e}" for n, x, y in zip(names, hist[k][s], hist[2][s])})
