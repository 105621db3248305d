#!/usr/bin/env python3
"""Within-process phase ablation of the contraction kernels at the benchmark
config (cdna_hip_programming.md: ablate before optimizing).  Prints the average
duration of fwd/bwd/apply/plan/pool kernels with phases skipped."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E, tt_embeddings_ops as ops

dev = torch.device("cuda:0")
cfg = dict(p=[200, 220, 250], q=[4, 4, 4], r=[32, 32])
if len(sys.argv) > 1 and sys.argv[1] == "cfg4":
    cfg = dict(p=[200, 220, 250], q=[4, 4, 8], r=[64, 64])
E_, D = int(np.prod(cfg["p"])), int(np.prod(cfg["q"]))
m = ops.TTEmbeddingBag(E_, D, cfg["r"], cfg["p"], cfg["q"], sparse=True, use_cache=False, weight_dist="uniform", device=dev)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 10, 512, 1, 20, E_)]
grad = torch.from_numpy(G.make_grad(2, 1, 512, D)[0]).to(dev)
names = ["fwd", "bwd", "apply", "plan", "pool"]

def run(mask, chunk=0, steps=30):
    E.debug_skip(mask)
    E.set_chunk(chunk)
    for k in range(5):
        m(*reqs[k % 10]).backward(grad)
    torch.cuda.synchronize()
    E.profile_reset(); E.profile_enable(0x1F)
    for k in range(steps):
        m(*reqs[k % 10]).backward(grad)
    torch.cuda.synchronize()
    E.profile_enable(0)
    out = {}
    for w, nm in enumerate(names):
        n, ms = E.profile_read(w)
        out[nm] = ms / max(n, 1) * 1e3
    E.debug_skip(0)
    return out

for chunk in (0, 8, 32):
    base = run(0, chunk)
    print(f"chunk={chunk}: " + "  ".join(f"{k}={v:.1f}us" for k, v in base.items()))
print("spec bwd cut points (16=after chunk_rec, 32=after records, 64=after all loads+B1 staged, 12=+LDS puts, 6=+GEMM1+tail, 7=no stores/no dB1 red., 1=no pc0/pc2 stores, 4=no dB1 reduction)")
for mask in (16, 32, 64, 8 | 4, 2 | 4, 1 | 2 | 4, 1, 4):
    r = run(mask)
    print(f"  mask={mask:2d}: bwd={r['bwd']:.1f}us fwd={r['fwd']:.1f}us")

print("non-temporal partial stores (512 = thin cores, 1024 = pivot): bwd + apply")
for mask in (0, 512, 1024, 1536, 0, 1536):
    r = run(mask, steps=100)
    print(f"  mask={mask:4d}: bwd={r['bwd']:.1f}us apply={r['apply']:.1f}us sum={r['bwd'] + r['apply']:.1f}us")
