"""Randomised parity sweeps, GPU (libttx through the C ABI shim) vs the CPU oracle.  TEST INFRASTRUCTURE.

run_plan_cases: geometry (T, q, ranks, p up to 3000), table count, batch, ragged / empty bags and a
70 %-on-three-indices skew drawn at random so that every plan route (tiny / single launch / wave units / wide
digit / table groups / multi-pass), the generic kernels' block walk of core 1 (small LDS budgets), per-table row factors (ttx_geom::p_tables) and the module's route
(ttx_lookup_prologue) are hit; forward + dense gradients.
run_cache_cases: the cache-live prologue (offsets -> bag rows, cache lookup, stable partition: bit-exact) and the
cache gather / SGD scatter.

A fixed-seed slice of each runs under `-m gpu` (tests/test_fuzz_gpu.py); scripts/fuzz_plan.py and
scripts/fuzz_cache.py run them for a time budget."""
import time

import numpy as np
import torch

import gen_inputs as G
import oracle_lib as O
from util import assert_close

dev = torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def run_plan_cases(seed=0, max_cases=None, budget=None):
    """-> (cases run, {route: count})"""
    import tt_embeddings as E

    rs = np.random.RandomState(seed)
    t0, n, routes = time.time(), 0, {}
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or n < max_cases):
        T = int(rs.choice([2, 3, 3, 3, 4]))
        tables = int(rs.choice([1, 1, 2, 3, 5, 9, 20, 40]))
        spec = T == 3 and rs.rand() < 0.4
        if spec:
            # the shape-specialised kernels' families: q0 = 4 / 2, q1 = 2 / 4 / 8 (ttx_tt_spec.inc)
            q = [[4, 4, 4], [4, 4, 8], [2, 4, 4], [2, 2, 4], [4, 8, 8]][int(rs.randint(0, 5))]
            r = [1] + [int(rs.choice([16, 32]))] * 2 + [1]
        else:
            q = [int(rs.randint(1, 5)) for _ in range(T)]
            r = [1] + [int(rs.randint(1, 9)) for _ in range(T - 1)] + [1]
            if rs.rand() < 0.3:  # larger ranks: K blocks of core 1 (r1 > 32) once the LDS budget below is small
                r = [1] + [int(rs.randint(1, 72)) for _ in range(T - 1)] + [1]
        # the generic kernels' block walk (K blocks x column passes of a core_1 slice, csrc/ttx_tt_generic.inc): a small
        # LDS budget drives these small shapes through it
        lds_kb = 0 if spec else int(rs.choice([0, 0, 12, 16, 24, 40]))
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
        E.debug_lds_budget(lds_kb * 1024)
        walk = E.debug_tiles(tables, p, q, r)
        if not spec and walk["MC"] == 0:  # (nothing fits that budget)
            E.debug_lds_budget(0)
            walk = E.debug_tiles(tables, p, q, r)
        if walk["ncp"] * walk["nkb"] > 1:
            routes["block-walk"] = routes.get("block-walk", 0) + 1
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
            tol = dict(rtol=2e-5, atol_scale=4e-6)  # (measured worst: 0.8x the default bound, profiles/r05_tolerances.md)
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
            E.debug_lds_budget(0)
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
        what = f"case {n}: T={T} tables={tables} p={p} q={q} r={r} B={B} nnz={nnz} walk={walk}"
        tol = dict(rtol=5e-5, atol_scale=1e-5)  # (hot slices: thousands of terms in an order of their own; measured worst: 1.5x the default bound)
        assert_close(out.cpu().numpy(), ref_out, what + " out", **tol)
        for k in range(T):
            assert_close(grads[k].cpu().numpy(), ref_g[k], what + f" grad{k}", **tol)
        S = tables * max(p)
        route = "tiny" if nnz <= 1024 and E_ <= 2**32 else ("single" if S <= 256 and nnz <= 16384 else ("units" if S <= 256 else ("wide" if S <= 2048 else "multi-pass")))
        routes[route] = routes.get(route, 0) + 1
        n += 1
        E.debug_lds_budget(0)
    return n, routes


def run_cache_cases(seed=0, max_cases=None, budget=None):
    """-> cases run"""
    import tt_embeddings as E

    rs = np.random.RandomState(seed)
    t0, n = time.time(), 0
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or n < max_cases):
        H = int(rs.choice([64, 4096, 1 << 16, 1 << 20]))
        E_ = int(rs.choice([50, 5000, 11_000_000]))
        a = float(rs.choice([1.05, 1.3, 2.0]))
        keys, freq = np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)
        for _ in range(int(rs.randint(1, 4))):
            O.update_cache_state((rs.zipf(a, size=int(rs.randint(1, 5000))) % E_).astype(np.int64), keys, freq)
        cs = int(rs.randint(1, 2000))
        state = np.where((keys != -1) & (rs.rand(H) < 0.6), rs.randint(0, cs, size=H), -1).astype(np.int32)
        nnz = int(rs.choice([1, 63, 64, 255, 256, 257, 1000, 4096, 20000, 70000]))
        B = int(rs.choice([1, 5, 64, 512, 3000]))
        B = max(B, nnz // 300)  # (a row hit m times in a bag takes m*g here and g m times in the oracle: keep m modest)
        lens = rs.multinomial(nnz, np.ones(B) / B)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        idx = (rs.zipf(a, size=nnz) % E_).astype(np.int64)
        exp = O.preprocess_indices(idx, off, 1, False, keys, state)
        got = E.preprocess_indices_sync(t(idx), t(off), 1, False, t(keys), t(state))
        what = f"case {n}: H={H} nnz={nnz} B={B}"
        assert got[3] == exp[3], what + " num_tt"
        ntt = exp[3]
        for k, name in ((0, "colidx"), (1, "rowidx"), (2, "tableidx")):
            assert np.array_equal(got[k].cpu().numpy(), exp[k]), what + " " + name
        if ntt < nnz:
            assert np.array_equal(got[4].cpu().numpy()[ntt:], exp[4][ntt:]), what + " cache locations"
            D = int(rs.choice([4, 64, 60, 128, 7]))
            loc, rowidx = exp[4][ntt:].astype(np.int32), exp[1][ntt:]
            w = rs.randn(cs, D).astype(np.float32)
            out0 = rs.randn(1, B, D).astype(np.float32)
            ref = out0.copy()
            O.cache_forward(B, loc, rowidx, w, ref[0])
            dout = t(out0)
            E.cache_forward(B, nnz - ntt, t(loc), t(rowidx), t(w), dout)
            assert_close(dout.cpu().numpy(), ref, what + f" cache_forward D={D}")
            grad = (rs.rand(B, D) * 0.1).astype(np.float32)
            # reference in float64 (a row can take tens of thousands of adds here: the fp32 oracle's own sequential
            # rounding is then larger than the GPU's, whose partial sums are shorter)
            w_ref = w.astype(np.float64)
            np.subtract.at(w_ref, loc, 0.1 * grad[rowidx].astype(np.float64))
            dw = t(w)
            E.cache_backward_sgd(nnz - ntt, t(grad), t(loc), t(rowidx), 0.1, dw)
            assert_close(dw.cpu().numpy(), w_ref, what + f" cache_backward_sgd D={D}", rtol=5e-5, atol_scale=1e-5)  # (float atomics in hardware order; measured worst: 0.2x the default bound)
        n += 1
    return n
