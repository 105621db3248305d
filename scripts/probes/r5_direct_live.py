"""diagnostic: cache-live route -- do two modules in the same state stay in the same state, step by step?"""
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
ops._DIRECT_BACKWARD = False
a, b = fresh(), fresh()
with torch.no_grad():
    for i, o, _ in batches: a(i, o)
a.cache_populate()
torch.cuda.synchronize()
b.load_state_dict(a.state_dict()); b.warmup = False
torch.cuda.synchronize()
def table(m):
    k, f, s = m.hashtbl.cpu().numpy(), m.cache_freq.cpu().numpy(), m.cache_state.cpu().numpy()
    return sorted(zip(k[k >= 0].tolist(), f[k >= 0].tolist(), s[k >= 0].tolist()))
print("start: cores equal", all(torch.equal(x, y) for x, y in zip(a.tt_cores, b.tt_cores)), "cache rows equal", torch.equal(a.cache_weight, b.cache_weight),
      "tables equal", table(a) == table(b), "keys", len(table(a)), "cached", sum(1 for x in table(a) if x[2] >= 0))
for k, (i, o, g) in enumerate(batches):
    na = E.preprocess_indices_sync(i, o, 1, False, a.hashtbl, a.cache_state)[3]
    nb = E.preprocess_indices_sync(i, o, 1, False, b.hashtbl, b.cache_state)[3]
    oa = a(i, o); ob = b(i, o)
    print(f"step {k}: misses a {na} b {nb} of {i.numel()}; outputs equal {torch.equal(oa, ob)} max diff {float((oa - ob).abs().max()):.3e}")
    oa.backward(g); ob.backward(g)
    torch.cuda.synchronize()
    print("   cores max diff", [float((x - y).abs().max()) for x, y in zip(a.tt_cores, b.tt_cores)], "cache rows", float((a.cache_weight - b.cache_weight).abs().max()),
          "tables equal", table(a) == table(b))
