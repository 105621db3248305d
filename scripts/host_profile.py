#!/usr/bin/env python3
"""Host-side cost of one eager module step at cfg2 (GPU box): wall time of the forward call, of the backward call, and a
cProfile of 300 steps.  The GPU work of a step is ~53 us; whatever the host needs beyond that is what an eager loop pays."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))
import torch
import tt_embeddings_ops as ops

dev = torch.device("cuda:0")
m = ops.TTEmbeddingBag(num_embeddings=11_000_000, embedding_dim=64, tt_ranks=[32, 32], tt_p_shapes=[200, 220, 250],
                       tt_q_shapes=[4, 4, 4], sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.01, use_cache=True,
                       cache_size=1024, hashtbl_size=1 << 20, weight_dist="uniform", device=dev)
B, L = 512, 20
g = torch.Generator().manual_seed(0)
batches = [(torch.randint(0, 11_000_000, (B * L,), generator=g).to(dev), torch.arange(0, B * L + 1, L).to(dev)) for _ in range(10)]
grad = torch.rand(B, 64, device=dev)


def step(i):
    idx, off = batches[i % 10]
    out = m(idx, off)
    out.backward(grad)


for i in range(50):
    step(i)
torch.cuda.synchronize()
N = 2000
tf = tb = 0.0
t0 = time.perf_counter()
for i in range(N):
    idx, off = batches[i % 10]
    a = time.perf_counter()
    out = m(idx, off)
    b = time.perf_counter()
    out.backward(grad)
    c = time.perf_counter()
    tf += b - a
    tb += c - b
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"eager step {1e6 * (t1 - t0) / N:.1f} us wall; host time in forward call {1e6 * tf / N:.1f} us, in backward call {1e6 * tb / N:.1f} us")
# the same with the queue drained every step (pure host latency of the calls, GPU idle at call time)
tf = tb = 0.0
for i in range(500):
    idx, off = batches[i % 10]
    torch.cuda.synchronize()
    a = time.perf_counter()
    out = m(idx, off)
    b = time.perf_counter()
    out.backward(grad)
    c = time.perf_counter()
    tf += b - a
    tb += c - b
print(f"with an empty queue: forward call {1e6 * tf / 500:.1f} us, backward call {1e6 * tb / 500:.1f} us")
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue())
