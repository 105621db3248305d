"""Does a hipGraph run independent branches concurrently on this stack?  Two chains of small single-work-group kernels
(the duplicate-map sort, ~50 us each, one work-group) captured (a) on one stream, (b) on two forked streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))
import torch
import tt_embeddings as E

dev = torch.device("cuda:0")
p, q, r = [200, 220, 250], [4, 4, 4], [1, 32, 32, 1]
idx = [torch.randint(0, 11_000_000, (10240,), device=dev) for _ in range(2)]
tb = torch.zeros(10240, dtype=torch.int64, device=dev)
for i in idx:
    E.make_plan(1, p, q, r, 10240, i, tb, None, dedup=True)
torch.cuda.synchronize()


def chain(i, n):
    keep = []
    for _ in range(n):
        keep.append(E.make_plan(1, p, q, r, 10240, i, tb, None, dedup=True))
    return keep


def timed(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    k1 = chain(idx[0], 8) + chain(idx[1], 8)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s1):
    s2.wait_stream(s1)
    a = chain(idx[0], 8)
    with torch.cuda.stream(s2):
        b = chain(idx[1], 8)
    s1.wait_stream(s2)
print(f"one stream, 16 launches pairs: {timed(g1):.1f} us per replay; two forked streams, 8 + 8: {timed(g2):.1f} us per replay")
# eager, two streams
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1):
        chain(idx[0], 8)
    with torch.cuda.stream(s2):
        chain(idx[1], 8)
torch.cuda.synchronize()
print(f"eager two streams: {(time.perf_counter() - t0) / 5 * 1e6:.1f} us per 8 + 8")
