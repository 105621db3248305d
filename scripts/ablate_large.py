#!/usr/bin/env python3
"""Phase ablation of the shape-specialised contraction kernels at a LARGE batch (default: cfg5's per-GPU shard, 4 tables x
4096 bags x 20 lookups = 327,680 lookups), where the kernels -- not the launches -- are the bound.  ttx_debug_skip cut points
leave phases of spec_bwd_kernel out (results invalid, timing only); HIP events around each launch.
    python scripts/ablate_large.py [tables] [B] [q0,q1,q2] [r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E, tt_embeddings_ops as ops

dev = torch.device("cuda:0")
tables = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
q = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 4, 4]
rk = int(sys.argv[4]) if len(sys.argv) > 4 else 32
p, r = [200, 220, 250], [rk, rk]
E_, D = int(np.prod(p)), int(np.prod(q))
kw = dict(sparse=True, use_cache=False, weight_dist="uniform", device=dev)
m = ops.TTEmbeddingBag(E_, D, r, p, q, **kw) if tables == 1 else ops.TableBatchedTTEmbeddingBag(tables, E_, D, r, p, q, **kw)
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 4, B, tables, 20, E_)]
g = G.make_grad(2, tables, B, D)
grad = torch.from_numpy(g[0] if tables == 1 else g).to(dev)
names = ["fwd", "bwd", "apply", "plan", "pool"]
nnz = tables * B * 20
fl = 2.0 * (q[0] * rk * q[1] * rk + q[0] * q[1] * rk * q[2])


def run(mask, steps=12):
    E.debug_skip(mask)
    for k in range(3):
        m(*reqs[k % 4]).backward(grad)
    torch.cuda.synchronize()
    E.profile_reset(); E.profile_enable(0x1F)
    for k in range(steps):
        m(*reqs[k % 4]).backward(grad)
    torch.cuda.synchronize()
    E.profile_enable(0)
    out = {}
    for w, nm in enumerate(names):
        n, ms = E.profile_read(w)
        out[nm] = ms / max(n, 1) * 1e3
    E.debug_skip(0)
    return out


base = run(0)
print(f"{tables} tables x {B} bags, q={q} r={rk}: {nnz} lookups  " + "  ".join(f"{k}={v:.1f}us" for k, v in base.items()))
print(f"  fwd {fl * nnz / base['fwd'] / 1e6:.1f} TF ({fl * nnz / base['fwd'] / 1e6 / 157.3:.3f}), bwd {2 * fl * nnz / base['bwd'] / 1e6:.1f} TF ({2 * fl * nnz / base['bwd'] / 1e6 / 157.3:.3f})")
print("spec bwd cut points (16=after chunk_rec, 32=after records, 64=after all loads+B1 staged, 12=+LDS puts, 6=+GEMM1+tail(+dX0 in LDS), "
      "4=+d core_0 GEMM (no d core_1), 2=GEMM1+tail+d core_1, 1=no pc0/pc2 stores, 128=no pivot partial store)")
for mask in (16, 32, 64, 8 | 4, 2 | 4, 4, 2, 1, 128, 1 | 128):
    rr = run(mask)
    print(f"  mask={mask:3d}: bwd={rr['bwd']:.1f}us")
