"""eager fwd+bwd(SGD) step over 8 tables of 4 different cardinalities (B=512, 20 lookups per bag): one
TTEmbeddingBag per table vs MixedTTEmbeddingBag grouped by TT shape (without / with a HIP stream per group)
vs fused (ONE batched lookup over all tables, ttx_geom::p_tables); the fused one also as a hipGraph round"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))
import tt_embeddings_ops as ops, ttx_mixed, ttx_graph

dev = torch.device("cuda:0")
D, q, r, B, L = 64, [4, 4, 4], [32, 32], 512, 20
shapes = {10_000_000: [200, 220, 250], 5_000_000: [160, 180, 200], 1_000_000: [100, 100, 100], 300_000: [64, 70, 72]}
Es = [10_000_000, 5_000_000, 1_000_000, 300_000] * 2
ps = [shapes[e] for e in Es]
kw = dict(sparse=True, optimizer=ops.OptimType.SGD, learning_rate=0.05, weight_dist="uniform", device=dev)
rs = np.random.RandomState(0)
reqs = []
for it in range(10):
    idx = [torch.from_numpy(rs.randint(0, e, size=B * L).astype(np.int64)).to(dev) for e in Es]
    off = [torch.arange(0, B * L, L, dtype=torch.int64, device=dev) for _ in Es]
    reqs.append((idx, off))
grads = [torch.rand(B, D, device=dev) * 0.1 for _ in Es]

def timeit(step, n=100):
    for k in range(10): step(*reqs[k % 10])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n): step(*reqs[k % 10])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

singles = [ops.TTEmbeddingBag(Es[k], D, r, ps[k], q, use_cache=False, include_last_offset=False, **kw) for k in range(len(Es))]
def step_single(idx, off):
    outs = [m(i, o) for m, i, o in zip(singles, idx, off)]
    torch.autograd.backward(outs, grads)
print(f"one module per table         : {timeit(step_single):.3f} ms/step")
for streams, fused in ((False, False), (True, False), (False, True)):
    mm = ttx_mixed.MixedTTEmbeddingBag(Es, D, r, ps, q, include_last_offset=False, streams=streams, fused=fused, **kw)
    def step_mixed(idx, off):
        torch.autograd.backward(mm(idx, off), grads)
    print(f"mixed, streams={streams!s:5} fused={fused!s:5} ({len(mm.groups)} groups): {timeit(step_mixed):.3f} ms/step")
rnd = ttx_graph.GraphedRound(step_mixed, reqs, warmup=2)
for _ in range(3): rnd.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): rnd.replay()
torch.cuda.synchronize()
print(f"fused, hipGraph round        : {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")

# ---- f4, the rest of the row: tables that differ in TT ranks / factoring -> one launch set per (q, ranks) group.  Eagerly the
# groups run one after the other (host-bound either way); captured into ONE hipGraph with a HIP stream per group they are
# parallel branches.  8 tables: r = 32 (x3), r = 16 (x3), r = 64 q = [4,4,4] (x1), r = 13/12 (generic kernels, x1).
ranks2 = [[32, 32], [16, 16], [32, 32], [16, 16], [64, 64], [32, 32], [16, 16], [13, 12]]
def launches(fn):
    import tt_embeddings as E
    E.profile_reset(); E.profile_enable(0x3F); fn(); torch.cuda.synchronize(); E.profile_enable(0)
    return sum(E.profile_read(w)[0] for w in range(6))
for streams, pad in ((False, False), (True, False), (False, True)):
    m2 = ttx_mixed.MixedTTEmbeddingBag(Es, D, ranks2, ps, q, include_last_offset=False, streams=streams, fused=True, pad_ranks=pad, **kw)
    def step2(idx, off):
        torch.autograd.backward(m2(idx, off), grads)
    ms = timeit(step2)
    n_l = launches(lambda: step2(*reqs[0]))
    r2 = ttx_graph.GraphedRound(step2, reqs, warmup=2)
    for _ in range(3): r2.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r2.replay()
    torch.cuda.synchronize()
    print(f"mixed ranks ({len(m2.groups)} launch sets, {n_l} kernel launches per step), streams={streams!s:5} pad_ranks={pad!s:5}: eager {ms:.3f} ms/step, "
          f"hipGraph round {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
# the same with ranks 32 / 16 only (padding 16 -> 32 costs 4x the multiply-adds of the small tables, not 16x)
ranks3 = [[32, 32], [16, 16]] * 4
for pad in (False, True):
    m3 = ttx_mixed.MixedTTEmbeddingBag(Es, D, ranks3, ps, q, include_last_offset=False, fused=True, pad_ranks=pad, **kw)
    def step3(idx, off):
        torch.autograd.backward(m3(idx, off), grads)
    ms = timeit(step3)
    n_l = launches(lambda: step3(*reqs[0]))
    r3 = ttx_graph.GraphedRound(step3, reqs, warmup=2)
    for _ in range(3): r3.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r3.replay()
    torch.cuda.synchronize()
    print(f"ranks 32 / 16 ({len(m3.groups)} launch sets, {n_l} kernel launches per step), pad_ranks={pad!s:5}: eager {ms:.3f} ms/step, "
          f"hipGraph round {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
# tables that differ in the factoring q of D = 64 (round 4): [4,4,4] x4 and [2,4,8] x4, ranks 32 -- two launch sets (streams: parallel
# branches of the graph) against ONE over q = [4,4,8] with every table's cores zero-padded to it (pad_q)
q4 = [[4, 4, 4], [2, 4, 8]] * 4
for streams, padq in ((False, False), (True, False), (False, True)):
    m4 = ttx_mixed.MixedTTEmbeddingBag(Es, D, r, ps, q4, include_last_offset=False, streams=streams, fused=True, pad_q=padq, **kw)
    def step4(idx, off):
        torch.autograd.backward(m4(idx, off), grads)
    ms = timeit(step4)
    n_l = launches(lambda: step4(*reqs[0]))
    r4 = ttx_graph.GraphedRound(step4, reqs, warmup=2)
    for _ in range(3): r4.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r4.replay()
    torch.cuda.synchronize()
    print(f"q = [4,4,4] / [2,4,8] ({len(m4.groups)} launch sets, {n_l} engine launches per step), streams={streams!s:5} pad_q={padq!s:5}: eager {ms:.3f} ms/step, "
          f"hipGraph round {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
