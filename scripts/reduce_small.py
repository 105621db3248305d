"""reduce+apply / backward duration vs the number of lookups (uniform and rank-ordered Zipf tails), SGD, cfg2 geometry"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fbtt-embedding_amd"))
import tt_embeddings as E

dev = torch.device("cuda:0")
p, q, r = [200, 220, 250], [4, 4, 4], [1, 32, 32, 1]
E_, D, B = 200 * 220 * 250, 64, 512
Lt = torch.tensor([220 * 250, 250, 1], dtype=torch.int64, device=dev)
cores = [torch.randn(1, p[k], r[k] * q[k] * r[k + 1], device=dev) * 0.1 for k in range(3)]
rs = np.random.RandomState(0)
PROF = {"fwd": 0, "bwd": 1, "reduce": 2, "plan": 3, "pool": 4}
CASES = (("uniform", 10240, "u"), ("uniform", 1230, "u"), ("tail", 1230, "t"), ("uniform", 300, "u"))
if len(sys.argv) > 1:
    CASES = (("uniform", int(sys.argv[1]), "u"),)
for label, nnz, kind in CASES:
    if kind == "u":
        idx = rs.randint(0, E_, size=nnz)
    else:  # ranks beyond 100k of a Zipf(1.2), rank == index
        u = rs.rand(nnz)
        a, b = 1e5 ** -0.2, 1.1e7 ** -0.2
        idx = ((a - u * (a - b)) ** (-5.0)).astype(np.int64) % E_
    idx = torch.from_numpy(idx.astype(np.int64)).to(dev)
    rowidx = torch.from_numpy(np.sort(rs.randint(0, B, size=nnz)).astype(np.int64)).to(dev)
    tableidx = torch.zeros(nnz, dtype=torch.int64, device=dev)
    d_out = torch.randn(1, B, D, device=dev)
    plan = E.make_plan(1, p, q, r, nnz, idx, tableidx, rowidx)
    for it in range(3):
        E.tt_sgd_backward(1000, D, 0.01, p, q, r, Lt, nnz, idx, rowidx, tableidx, d_out, cores, plan=plan)
    torch.cuda.synchronize()
    E.profile_enable(0xff)
    E.profile_reset()
    for it in range(30):
        E.tt_sgd_backward(1000, D, 0.01, p, q, r, Lt, nnz, idx, rowidx, tableidx, d_out, cores, plan=plan)
    torch.cuda.synchronize()
    out = {k: E.profile_read(w) for k, w in PROF.items()}
    E.profile_enable(0)
    print(label, nnz, {k: round(ms / max(n, 1) * 1e3, 2) for k, (n, ms) in out.items() if n})
