"""randomised parity sweep (GPU vs CPU oracle): geometry, table count, batch, bag sizes and skew drawn at random so
that every plan route (tiny / single launch / wave units / wide digit / multi-pass) is hit; forward + dense grads.
usage: python scripts/fuzz_plan.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fbtt-embedding_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import gen_inputs as G, oracle_lib as O, tt_embeddings as E
from util import assert_close

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n, routes = time.time(), 0, {}
while time.time() - t0 < budget:
    T = int(rs.choice([2, 3, 3, 3, 4]))
    tables = int(rs.choice([1, 1, 2, 3, 5, 9, 20, 40]))
    spec = T == 3 and rs.rand() < 0.4
    if spec:
        q, r = [4, 4, int(rs.choice([4, 8]))], [1] + [int(rs.choice([16, 32]))] * 2 + [1]
        r[2] = r[1]
    else:
        q = [int(rs.randint(1, 5)) for _ in range(T)]
        r = [1] + [int(rs.randint(1, 9)) for _ in range(T - 1)] + [1]
    pmax = int(rs.choice([6, 40, 300, 700, 3000]))
    p = [int(rs.randint(2, pmax + 1)) for _ in range(T)]
    if np.prod(np.array(p, dtype=np.float64)) * 1.0 > 2e12:
        continue
    E_ = int(np.prod(np.array(p, dtype=np.int64)))
    D = int(np.prod(q))
    B = int(rs.choice([1, 7, 64, 300, 1500]))
    pf = int(rs.choice([1, 3, 10, 40]))
    nnz_est = tables * B * pf
    if nnz_est > 150000:
        continue
    lens = rs.randint(0, 2 * pf + 1, size=tables * B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(off[-1])
    if nnz == 0:
        continue
    idx = rs.randint(0, E_, size=nnz).astype(np.int64)
    if rs.rand() < 0.3:
        hot = rs.randint(0, E_, size=3)
        idx = np.where(rs.rand(nnz) < 0.7, hot[rs.randint(0, 3, size=nnz)], idx).astype(np.int64)
    d_out = G.make_grad(int(rs.randint(1 << 30)), tables, B, D)
    if tables > 1 and rs.rand() < 0.35:
        # tables of different row factors (ttx_geom::p_tables): the oracle does every table on its own
        ps = [[int(rs.randint(2, pmax + 1)) for _ in range(T)] for _ in range(tables)]
        Es = [int(np.prod(np.array(pk, dtype=np.int64))) for pk in ps]
        if max(Es) > 2e12:
            continue
        bounds = off[::B]
        idx = np.concatenate([rs.randint(0, Es[k], size=int(bounds[k + 1] - bounds[k])) for k in range(tables)]).astype(np.int64)
        tcores = [G.make_cores(int(rs.randint(1 << 30)), 1, ps[k], q, r[1:-1], "signed") for k in range(tables)]
        gc = [t(np.concatenate([tcores[k][c_] for k in range(tables)], axis=1)) for c_ in range(T)]
        Lt = torch.zeros(T, dtype=torch.int64, device=dev)
        if rs.rand() < 0.5:  # the module's route: offsets -> rows + plan (table groups when the slice ids need them)
            ri, ti, plan = E.lookup_prologue(t(idx), t(off), tables, ps, q, r)
            routes["prologue"] = routes.get("prologue", 0) + 1
        else:
            ri, ti = E.preprocess_indices_sync(t(idx), t(off), tables, True, torch.empty(0, dtype=torch.int64, device=dev),
                                               torch.empty(0, dtype=torch.int32, device=dev))[1:3]
            plan = E.make_plan(tables, ps, q, r, nnz, t(idx), ti, ri)
        out = E.tt_forward(1000, tables, B, D, ps, q, r, Lt, nnz, t(idx), ri, ti, gc, plan=plan)
        grads = E.tt_dense_backward(1000, D, ps, q, r, Lt, nnz, t(idx), ri, ti, t(d_out), gc, plan=plan)
        what = f"case {n} (mixed): T={T} p={ps} q={q} r={r} B={B} nnz={nnz}"
        tol = dict(rtol=1e-4, atol_scale=2e-5)
        gsplit = [torch.split(grads[c_][0], [pk[c_] for pk in ps], dim=0) for c_ in range(T)]
        for k in range(tables):
            gk = O.make_geom(1, ps[k], q, r)
            ik, ok = idx[bounds[k]:bounds[k + 1]], off[k * B:(k + 1) * B + 1] - bounds[k]
            rk, tk = O.rowidx_from_offsets(ok, 1)
            assert_close(out[k].cpu().numpy(), O.tt_forward(gk, B, D, ik, rk, tk, tcores[k])[0], what + f" out table {k}", **tol)
            rg = O.tt_backward(gk, O.OPTIM_DENSE, B, D, 0, 0, ik, rk, tk, d_out[k:k + 1], [x.copy() for x in tcores[k]])
            for c_ in range(T):
                assert_close(gsplit[c_][k].cpu().numpy(), rg[c_][0], what + f" grad{c_} table {k}", **tol)
        routes["mixed"] = routes.get("mixed", 0) + 1
        n += 1
        continue
    cores = G.make_cores(int(rs.randint(1 << 30)), tables, p, q, r[1:-1], "signed")
    c = dict(tables=tables, T=T, p=p, q=q, r=r, B=B, D=D)
    g = O.make_geom(tables, p, q, r)
    rowidx, tableidx = O.rowidx_from_offsets(off, tables)
    ref_out = O.tt_forward(g, B, D, idx, rowidx, tableidx, cores)
    ref_g = O.tt_backward(g, O.OPTIM_DENSE, B, D, 0, 0, idx, rowidx, tableidx, d_out, [x.copy() for x in cores])
    Lt = t(np.array([int(np.prod(p[k + 1:])) for k in range(T)], dtype=np.int64))
    gc = [t(x) for x in cores]
    if tables > 1 and rs.rand() < 0.5:
        ri, ti, plan = E.lookup_prologue(t(idx), t(off), tables, p, q, r)
        routes["prologue"] = routes.get("prologue", 0) + 1
    else:
        ri, ti = E.preprocess_indices_sync(t(idx), t(off), tables, True, torch.empty(0, dtype=torch.int64, device=dev),
                                           torch.empty(0, dtype=torch.int32, device=dev))[1:3]
        plan = E.make_plan(tables, p, q, r, nnz, t(idx), ti, ri)
    out = E.tt_forward(1000, tables, B, D, p, q, r, Lt, nnz, t(idx), ri, ti, gc, plan=plan)
    grads = E.tt_dense_backward(1000, D, p, q, r, Lt, nnz, t(idx), ri, ti, t(d_out), gc, plan=plan)
    what = f"case {n}: T={T} tables={tables} p={p} q={q} r={r} B={B} nnz={nnz}"
    tol = dict(rtol=1e-4, atol_scale=2e-5)  # (hot slices: thousands of terms in an order of their own)
    assert_close(out.cpu().numpy(), ref_out, what + " out", **tol)
    for k in range(T):
        assert_close(grads[k].cpu().numpy(), ref_g[k], what + f" grad{k}", **tol)
    S = tables * max(p)
    route = "tiny" if nnz <= 1024 and E_ <= 2**32 else ("single" if S <= 256 and nnz <= 16384 else ("units" if S <= 256 else ("wide" if S <= 2048 else "multi-pass")))
    routes[route] = routes.get(route, 0) + 1
    n += 1
print(f"{n} cases ok in {time.time() - t0:.0f} s; routes {routes}")
