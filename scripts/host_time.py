"""where the eager step's host time goes (cfg2): wall time of the module call and of backward() with the GPU kept out of the
way (a synchronize before each segment: what is measured is enqueue time, not kernel time)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tt_embeddings_ops as ops, gen_inputs as G
dev = torch.device("cuda:0")
p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
E_, D, B, L = 11_000_000, 64, 512, 20
m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.1, use_cache=False, weight_dist="uniform", device=dev)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 10, B, 1, L, E_)]
grad = torch.rand(B, D, device=dev)
for k in range(50): m(*reqs[k % 10]).backward(grad)
torch.cuda.synchronize()
N = 300
tf = tb = 0.0
for k in range(N):
    i, o = reqs[k % 10]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m(i, o)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out.backward(grad)
    t3 = time.perf_counter()
    tf += t1 - t0; tb += t3 - t2
print(f"host enqueue time: forward {tf / N * 1e6:.1f} us, backward {tb / N * 1e6:.1f} us per step")
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(N): m(*reqs[k % 10]).backward(grad)
torch.cuda.synchronize()
print(f"free-running eager step: {(time.perf_counter() - t0) / N * 1e6:.1f} us")
# the pieces of the forward call
fast = ops._native_node()
i, o = reqs[0]
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    el = time.perf_counter() - t0; torch.cuda.synchronize(); return el / n * 1e6
with torch.no_grad():
    print(f"module call under no_grad (no autograd node): {t(lambda: m(i, o), 500):.1f} us")
print(f"torch.empty(512, 64): {t(lambda: torch.empty(512, 64, device=dev)):.2f} us")
x = torch.rand(512, 64, device=dev, requires_grad=True)
print(f"a trivial op + backward (x * 2).backward(grad): {t(lambda: (x * 2).backward(grad), 500):.1f} us")
