#!/usr/bin/env python3
"""cProfile of the eager module step at cfg2 (host-side overhead of the ctypes path)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E, tt_embeddings_ops as ops
dev = torch.device("cuda:0")
p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
E_, D = int(np.prod(p)), 64
m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, use_cache=True, weight_dist="uniform", device=dev)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 10, 512, 1, 20, E_)]
grad = torch.from_numpy(G.make_grad(2, 1, 512, D)[0]).to(dev)
def loop(n):
    for k in range(n):
        m(*reqs[k % 10]).backward(grad)
    torch.cuda.synchronize()
loop(50)
t0 = time.perf_counter(); loop(500); t1 = time.perf_counter()
print(f"eager: {(t1 - t0) / 500 * 1e6:.1f} us/step")
pr = cProfile.Profile(); pr.enable(); loop(500); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
