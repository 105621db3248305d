"""plain vs DedupTTEmbeddingBag, eager fwd+bwd(SGD) ms/step at the benchmark geometry, Zipf(1.2) and uniform
streams, no cache (the regime before cache_populate): python scripts/bench_dedup.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))
import tt_embeddings_ops as ops, ttx_dedup

dev = torch.device("cuda:0")
p, q, r, D, L = [200, 220, 250], [4, 4, 4], [32, 32], 64, 20
E_ = 11_000_000
kw = dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False, weight_dist="uniform", device=dev)
rs = np.random.RandomState(0)
for B in (512, 16384):
    for name, gen in (("zipf1.2", lambda n: (rs.zipf(1.2, size=n) - 1) % E_), ("uniform", lambda n: rs.randint(0, E_, size=n))):
        reqs = [(torch.from_numpy(gen(B * L).astype(np.int64)).to(dev), torch.arange(0, B * L + 1, L, dtype=torch.int64, device=dev))
                for _ in range(10)]
        grad = torch.rand(B, D, device=dev) * 0.1
        res = {}
        for kind in ("plain", "dedup"):
            m = ops.TTEmbeddingBag(E_, D, r, p, q, **kw)
            mod = ttx_dedup.DedupTTEmbeddingBag(m) if kind == "dedup" else m
            step = lambda i, o: mod(i, o).backward(grad)
            for k in range(10): step(*reqs[k])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(50): step(*reqs[k % 10])
            torch.cuda.synchronize(); res[kind] = (time.perf_counter() - t0) / 50 * 1e3
        u = int(torch.unique(reqs[0][0]).numel())
        print(f"B={B:6d} {name:8s} distinct {u}/{B * L}: plain {res['plain']:.3f} ms/step, dedup {res['dedup']:.3f} ms/step")
