"""probe: the eager loop with each request's prologue on the side stream, NOT waiting for the main stream (the request
tensors were seen before: an event from then covers their producers) -- against the plain loop"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tt_embeddings_ops as ops, gen_inputs as G
dev = torch.device("cuda:0")
p, q, r = [200, 220, 250], [4, 4, 4], [32, 32]
E_, D, B, L = 11_000_000, 64, 512, 20
m = ops.TTEmbeddingBag(E_, D, r, p, q, sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.1, use_cache=True, weight_dist="uniform", device=dev)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 10, B, 1, L, E_)]
grad = torch.rand(B, D, device=dev)
fast = ops._native_node()
side = m.prefetch_stream(dev)
def plain(i, o): m(i, o).backward(grad)
def overlapped(i, o):
    with torch.cuda.stream(side):
        pre = fast.prologue(i, o, 1, m.tt_p_shapes, m.tt_q_shapes, m.tt_ranks, m.hashtbl, m.cache_freq)
        done = torch.cuda.Event(); done.record(side)
    m._prefetched[(id(i), id(o))] = (i, o, tuple(pre), done, i, o, i._version, o._version, False)
    m(i, o).backward(grad)
def run(fn, n=2000):
    for k in range(100): fn(*reqs[k % 10])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n): fn(*reqs[k % 10])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for rep in range(3):
    print(f"plain {run(plain):.1f} us/step | prologue on the side stream {run(overlapped):.1f} us/step", flush=True)
