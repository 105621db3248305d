#!/usr/bin/env python3
"""Cut-point timing of the backward contraction kernel at cfg2: run under rocprofv3 --kernel-trace --stats with one
ablation mask per process (ttx_debug_skip: 16 = after the chunk record, 64 = after all loads, 12 = + LDS puts,
6 = + GEMM1 and tail, 4 = all but the d core_1 stage, 1 = no thin-core stores, 0 = full kernel).
usage: rocprofv3 --kernel-trace --stats -d out -- python scripts/bwd_cut_points.py <mask>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gen_inputs as G, tt_embeddings as E
dev = torch.device("cuda:0")
mask = int(sys.argv[1])
p, q, r = [200, 220, 250], [4, 4, 4], [1, 32, 32, 1]
reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in G.make_requests(1, 10, 512, 1, 20, 11_000_000)]
cores = [torch.from_numpy(c).to(dev) for c in G.make_cores(2, 1, p, q, r[1:-1])]
grad = torch.from_numpy(G.make_grad(3, 1, 512, 64)).to(dev)
Lt = torch.tensor([55000, 250, 1], dtype=torch.int64, device=dev)
plans = [E.lookup_prologue(i, o, 1, p, q, r) for i, o in reqs]
E.debug_skip(mask)
for k in range(300):
    i, o = reqs[k % 10]; row, tab, plan = plans[k % 10]
    E.tt_sgd_backward(1000, 64, 0.0, p, q, r, Lt, i.numel(), i, row, tab, grad, cores, plan=plan)
torch.cuda.synchronize()
