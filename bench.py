#!/usr/bin/env python3
"""bench.py -- fwd+bwd throughput of the TT-EmbeddingBag hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward + backward (fused SGD) of the module over one batch of
synthetic lookups already resident in HBM, i.e. what the reference's own
benchmark times (tt_embeddings_benchmark.py:183-187).

N = 1 : BASELINE.json configs[1], the repo benchmark config (E=11M, D=64,
        p=[200,220,250], q=[4,4,4], ranks=[32,32], B=512, L=20 -> nnz=10240,
        sparse SGD, use_cache=True but never populated -- exactly what the
        reference benchmark instantiates, :166-175 -- so every step also runs
        the hash-table frequency update).
N > 1 : N such tables, one per rank (table-sharded, ttx_sharded.py), the 512-bag
        batch split across ranks, RCCL all-to-all of indices in / pooled vectors
        out.  Per-GPU lookups stay 10240 per step ("weak").

value = true algorithmic GFLOP/s = 3 * 2*(q0 r1 q1 r2 + q0 q1 r2 q2) * nnz / time
(the reference's formula, :154-158/:190, WITHOUT its x iters slip; the README's
2657.6 "GFLOPS" is 265.8 on this scale -- BASELINE.md).  Printed as ONE JSON line
by rank 0, with the roofline of the dominant kernel (backward contraction,
timed live with HIP events on its stream) and a CPU baseline (the oracle, a
scalar port, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

P_SHAPES, Q_SHAPES, RANKS = [200, 220, 250], [4, 4, 4], [32, 32]
B_GLOBAL, POOL = 512, 20
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 MFMA/VALU dense peak (MI355X_MICROARCH.md)

# --workload: cfg2 is the bench line (BASELINE.json configs[1]); the others are the rest of
# SURVEY.md section 8(d)'s measurement list, for profiles/ -- never the driver's default.
WORKLOADS = {
    "cfg2": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "cfg3": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.2, populate=True),
    # cfg3's index stream before the cache is populated: every hot row goes through the contraction
    "cfg3warm": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.2, populate=False),
    "cfg4": dict(q=[4, 4, 8], ranks=[64, 64], tables=1, B=512, optimizer="adagrad", alpha=1.0, populate=False),
    # one rank's share of cfg5 at 8 GPUs: 4 of the 26 tables, the whole 4096-bag batch
    # shapes outside the specialised family (reference-default q for D = 32 has q0 = 2): generic kernels
    "d32": dict(q=[2, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d32q4": dict(q=[4, 2, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d128r32": dict(q=[4, 4, 8], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # all 26 tables of cfg5 on ONE GPU (2.13 M lookups per step): the table-batched path at scale
    "cfg5full": dict(q=[4, 4, 4], ranks=[32, 32], tables=26, B=4096, optimizer="sgd", alpha=1.0, populate=False),
    "tb16": dict(q=[4, 4, 4], ranks=[32, 32], tables=16, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "tb8": dict(q=[4, 4, 4], ranks=[32, 32], tables=8, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "tb4": dict(q=[4, 4, 4], ranks=[32, 32], tables=4, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "cfg5shard": dict(q=[4, 4, 4], ranks=[32, 32], tables=4, B=4096, optimizer="sgd", alpha=1.0, populate=False),
}


def flop_per_nnz_fwd(q, r):
    return 2.0 * (q[0] * r[0] * q[1] * r[1] + q[0] * q[1] * r[1] * q[2])


def dense_baseline(E_, D, reqs, grad, steps, warmup):
    """the table the TT cores replace: nn.EmbeddingBag(E, D, mode="sum", sparse=True) + SGD(lr=0.1) on the same
    requests, eager, as tt_embeddings_benchmark.py:195-211 does behind --run-baseline"""
    import torch

    dev = grad.device
    emb = torch.nn.EmbeddingBag(E_, D, mode="sum", sparse=True, include_last_offset=True, device=dev)
    opt = torch.optim.SGD(emb.parameters(), lr=0.1)
    g2 = grad.reshape(-1, D)

    def one(idx, off):
        opt.zero_grad(set_to_none=True)
        emb(idx, off).backward(g2)
        opt.step()

    for k in range(warmup):
        one(*reqs[k % len(reqs)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(*reqs[k % len(reqs)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": round(ms, 4), "table_bytes": E_ * D * 4, "what": "nn.EmbeddingBag(sparse=True) fwd+bwd + SGD step, eager"}


def cpu_baseline(requests, cores, d_out, budget_s=12.0):
    """the oracle (scalar C port, 1 thread) on the same requests: fwd + fused-SGD bwd"""
    import oracle_lib as O

    g = O.make_geom(1, P_SHAPES, Q_SHAPES, RANKS)
    D = int(np.prod(Q_SHAPES))
    cores = [c.copy() for c in cores]
    done, t0 = 0, time.perf_counter()
    nnz = 0
    while True:
        idx, off = requests[done % len(requests)]
        rowidx, tableidx = O.rowidx_from_offsets(off, 1)
        O.tt_forward(g, B_GLOBAL, D, idx, rowidx, tableidx, cores)
        O.tt_backward(g, O.OPTIM_SGD, B_GLOBAL, D, 0.1, 0.0, idx, rowidx, tableidx, d_out, cores)
        done += 1
        nnz += idx.size
        el = time.perf_counter() - t0
        if el > budget_s or done >= 64:
            break
    gflops = 3.0 * flop_per_nnz_fwd(Q_SHAPES, RANKS) * nnz / el / 1e9
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"{done} fwd+bwd(SGD) steps of the same requests ({nnz} lookups) in {el:.1f} s, "
                      f"oracle/ttx_oracle.c single thread; {el / nnz * 1e6:.2f} us/nnz"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cache", action="store_true", help="use_cache=False (skip the hash-table frequency update)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time the eager Python path only (no hipGraph replay)")
    ap.add_argument("--optimizer", default=None, choices=["sgd", "adagrad"])
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--run-baseline", action="store_true",
                    help="also time the uncompressed nn.EmbeddingBag(E, D, sparse) + SGD on the same requests "
                         "(tt_embeddings_benchmark.py --run-baseline; needs E*D*4 bytes per table)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="(test) take the N > 1 code path -- process group, sharded module, all-to-all -- with one rank")
    args = ap.parse_args()
    global Q_SHAPES, RANKS, B_GLOBAL
    wl = WORKLOADS[args.workload]
    Q_SHAPES, RANKS, B_GLOBAL = wl["q"], wl["ranks"], wl["B"]
    if args.optimizer is None:
        args.optimizer = wl["optimizer"]
    ntab = wl["tables"]
    if args.gpus > 1 and args.workload != "cfg2":
        raise SystemExit("--gpus N > 1 runs the cfg2-per-rank sharded workload only")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = world > 1 or args.force_sharded
    if args.force_sharded:
        os.environ["TTX_FORCE_EXCHANGE"] = "1"
        os.environ.setdefault("MASTER_PORT", "29561")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL brings up 128 channels per communicator by default on this GPU; in the sandboxed boxes of this
        # pool that alone took minutes (measured: > 120 s at world size 1, 3.6 s with 4 channels).  The two
        # exchanges of a step are < 1 MB per peer: a handful of channels carries them.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        dist.init_process_group("nccl", device_id=dev)

    import gen_inputs as G
    import tt_embeddings as E
    import tt_embeddings_ops as ops
    import ttx_sharded

    E_, D = int(np.prod(P_SHAPES)), int(np.prod(Q_SHAPES))
    opt = ops.OptimType.SGD if args.optimizer == "sgd" else ops.OptimType.EXACT_ADAGRAD
    iters = 10  # request batches, like the reference's --iters
    hit_rate = None
    B_local = B_GLOBAL // world
    assert B_local * world == B_GLOBAL
    torch.manual_seed(1234 + rank)
    if not sharded:
        use_cache = (not args.no_cache) and ntab == 1
        kw = dict(sparse=True, optimizer=opt, learning_rate=0.1, use_cache=use_cache, weight_dist="uniform", device=dev)
        if wl["populate"]:  # cfg3: 256Ki-row cache behind a 1Mi-slot table
            kw.update(cache_size=1 << 18, hashtbl_size=1 << 20)
        if ntab == 1:
            mod = ops.TTEmbeddingBag(E_, D, RANKS, P_SHAPES, Q_SHAPES, **kw)
        else:
            mod = ops.TableBatchedTTEmbeddingBag(ntab, E_, D, RANKS, P_SHAPES, Q_SHAPES, **kw)
        cores_np = G.make_cores(1234, ntab, P_SHAPES, Q_SHAPES, RANKS, "uniform")
        with torch.no_grad():
            for dst, src in zip(mod.tt_cores, cores_np):
                dst.copy_(torch.from_numpy(src))
        reqs_np = G.make_requests(1235, iters, B_GLOBAL, ntab, POOL, E_, alpha=wl["alpha"])
        reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in reqs_np]
        d_out_np = G.make_grad(1236, ntab, B_GLOBAL, D)
        grad = torch.from_numpy(d_out_np[0] if ntab == 1 else d_out_np).to(dev)
        step = lambda i, o: mod(i, o).backward(grad)  # noqa: E731
        nnz_step_total = ntab * B_GLOBAL * POOL
        if wl["populate"]:
            # warm the frequency table on a DIFFERENT request stream (same distribution), then populate:
            # the timed batches then mix cache hits (hot rows) with TT lookups (the Zipf tail)
            warm = G.make_requests(4321, 5 * iters, B_GLOBAL, ntab, POOL, E_, alpha=wl["alpha"])
            for i, o in warm:
                step(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev))
            mod.cache_populate()
            n_tt = sum(E.preprocess_indices_sync(i, o, 1, False, mod.hashtbl, mod.cache_state)[3] for i, o in reqs)
            hit_rate = 1.0 - n_tt / float(iters * nnz_step_total)
    else:
        mod = ttx_sharded.ShardedTableBatchedTTEmbeddingBag(
            world, E_, D, RANKS, tt_p_shapes=P_SHAPES, tt_q_shapes=Q_SHAPES, sparse=True, optimizer=opt,
            learning_rate=0.1, use_cache=False, weight_dist="uniform", device=dev)
        reqs_np = G.make_requests(1235 + rank, iters, B_local, world, POOL, E_)
        reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in reqs_np]
        grad = torch.from_numpy(G.make_grad(1236 + rank, world, B_local, D)).to(dev)
        step = lambda i, o: mod(i, o, fixed_pooling=POOL).backward(grad)  # noqa: E731
        nnz_step_total = world * B_GLOBAL * POOL  # every table sees the whole 512-bag batch

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(*reqs[k % iters])
    sync()

    # ---- region 1 (eager Python path): live HIP-event timing of the dominant kernel ----
    E.profile_reset()
    E.profile_enable(1 << E.PROF_BWD)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(*reqs[k % iters])
    sync()
    t1 = time.perf_counter()
    E.profile_enable(0)
    eager_elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(eager_elapsed, op=dist.ReduceOp.MAX)
    eager_elapsed = float(eager_elapsed.item())
    n_bwd, ms_bwd = E.profile_read(E.PROF_BWD)
    bwd_src = "HIP events around each launch of the eager region"

    # ---- region 2 (the reported value): the same fwd+bwd step captured once per request
    # batch into a hipGraph (HIP streams and graphs instead of per-launch host work) and
    # replayed; every replay runs the full plan/forward/pool/backward/apply kernel sequence
    # on inputs resident in HBM.  Falls back to the eager timing if capture is unavailable.
    mode, elapsed = "eager", eager_elapsed
    # (cache live: capturable only through the C++ node, which keeps the partition's split point on the device;
    # the reference-shaped Python route reads it back to the host every step)
    if not args.no_graph and not sharded and (not wl["populate"] or ops._native_node() is not None):
        try:
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                for k in range(3):
                    step(*reqs[k % iters])
            torch.cuda.current_stream().wait_stream(cap)
            torch.cuda.synchronize()
            graphs = []
            for i, o in reqs:
                E._ws_cache.clear()  # every graph owns its workspace (allocated from its pool)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    step(i, o)
                graphs.append(g)
            # one more graph holding a whole round of the request batches (iters steps): a graph launch costs
            # the host tens of microseconds here, as much as a step's kernels take, so the timed loop replays
            # rounds and falls back to the single-step graphs only for the remainder -- exactly K steps either way
            E._ws_cache.clear()
            g_round = torch.cuda.CUDAGraph()
            E.profile_reset()
            E.profile_mask(1 << E.PROF_BWD)  # the event pairs around the backward kernel become graph nodes
            with torch.cuda.graph(g_round, stream=cap, capture_error_mode="thread_local"):
                for i, o in reqs:
                    step(i, o)
            E.profile_mask(0)
            E._ws_cache.clear()

            def run_steps(n):
                for _ in range(n // iters):
                    g_round.replay()
                for k in range(n % iters):
                    graphs[k].replay()

            run_steps(max(args.warmup, iters))
            sync()
            t0 = time.perf_counter()
            run_steps(args.steps)
            sync()
            t1 = time.perf_counter()
            mode, elapsed = "hipgraph", t1 - t0
            # the captured event pairs now hold the times of the LAST replay of each of the round's steps: the
            # live duration of the dominant kernel inside the timed region, without the host's launch latency
            # that an eager event bracket picks up when the host, not the GPU, is the bottleneck
            try:
                n_g, ms_g = E.profile_read(E.PROF_BWD)
                if n_g > 0 and ms_g > 0:
                    n_bwd, ms_bwd, bwd_src = n_g, ms_g, "HIP events captured in the replayed graph"
            except RuntimeError:
                pass
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] graph capture unavailable ({type(ex).__name__}: {ex}); reporting the eager path", file=sys.stderr)
            torch.cuda.synchronize()

    def build_line(mode, elapsed, breakdown, a2a, note=None):
        fl_fwd = flop_per_nnz_fwd(Q_SHAPES, RANKS)
        ms_per_step = elapsed / args.steps * 1e3
        gflops = 3.0 * fl_fwd * nnz_step_total / (elapsed / args.steps) / 1e9
        per_rank_nnz = nnz_step_total // world
        # dominant kernel = backward contraction: 2/3 of the algorithmic fwd+bwd FLOP
        bwd_flop_per_launch = 2.0 * fl_fwd * per_rank_nnz
        bwd_us = ms_bwd / max(n_bwd, 1) * 1e3
        achieved = bwd_flop_per_launch / (bwd_us * 1e-6) / 1e12 if n_bwd else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_bwd_bytes.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": ("fwd+bwd GFLOPS (true algorithmic: 3 x fwd FLOP / time), TT-EmbeddingBag E=11M "
                       f"D={D} ranks={RANKS} nnz={per_rank_nnz}"),
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {'TTEmbeddingBag' if ntab == 1 else f'TableBatchedTTEmbeddingBag x{ntab} tables'} "
                                    f"E=11000000 D={D} p=[200,220,250] q={Q_SHAPES} ranks={RANKS} B={B_GLOBAL} L=20 "
                                    f"nnz={per_rank_nnz} sparse {args.optimizer.upper()}, use_cache="
                                    + ("False" if (args.no_cache or sharded or ntab > 1) else
                                       (f"True(populated from 50 other batches, 256Ki rows, Zipf a={wl['alpha']}, hit rate {hit_rate:.3f})" if wl["populate"] else "True(unpopulated)"))
                                    + ("" if not sharded else f"; {world} such tables, one per rank, table-sharded, RCCL all-to-all; B_local={B_local}")),
                       "nnz_per_step_total": nnz_step_total, "flop_per_nnz_fwd_bwd": 3.0 * fl_fwd,
                       "path": "Python module -> C++ autograd node (or ctypes) -> C ABI -> HIP" + (("; timed as hipGraph replay of the captured module fwd+bwd steps (one graph per round of the 10 request batches)"
                                + ("; all-to-all exchanges issued on RCCL directly, inside the graph" if "rccl" in mode else "")) if mode.startswith("hipgraph") else "; eager")},
            "timed_mode": mode,
            "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4),
            "eager_value": round(3.0 * flop_per_nnz_fwd(Q_SHAPES, RANKS) * nnz_step_total / (eager_elapsed / args.steps) / 1e9, 2),
            "us_per_nnz": round(elapsed / args.steps / nnz_step_total * 1e6, 5),
            "ref_formula_gflops_x_iters": round(gflops * 10, 1),
            "reference_readme_true_gflops": 265.8,
            "kernel_us": breakdown,
            "roofline": {"bound": "mfma", "kernel": "spec_bwd_kernel / bwd_kernel (backward contraction)", "achieved": round(achieved, 3),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 5),
                         "traffic": traffic, "launches": n_bwd, "avg_us": round(bwd_us, 2), "timed_by": bwd_src,
                         "flop_per_launch": bwd_flop_per_launch},
        }
        if a2a is not None:
            line["all_to_all"] = a2a
        if note:
            line["note"] = note
        return line


    # ---- region 2 for N > 1: the sharded step with its exchanges issued on RCCL directly (ttx_sharded.DirectExchange:
    # current stream, no side-stream hops) and captured, a round of request batches per graph.  torch.distributed's
    # own collectives cannot be captured on this stack.  Never validated on more than one rank when written, so a
    # watchdog thread prints the eager result and ends the process if this region does not finish in time.
    if not args.no_graph and sharded and ops._native_node() is not None and not os.environ.get("TTX_NO_DIRECT_RCCL"):
        import threading

        def bail():
            if rank == 0:
                print(json.dumps(build_line("eager", eager_elapsed, {}, None,
                                            note="direct-RCCL graph region did not finish; eager torch.distributed result")),
                      flush=True)
            os._exit(0)

        dog = threading.Timer(float(os.environ.get("TTX_DIRECT_TIMEOUT", "150")), bail)
        dog.daemon = True
        dog.start()
        try:
            import ttx_graph

            mod.enable_direct_exchange()
            for k in range(5):
                step(*reqs[k % iters])
            sync()
            rnd = ttx_graph.GraphedRound(step, reqs, warmup=2)
            for _ in range(max(1, args.warmup // iters)):
                rnd.replay()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps // iters):
                rnd.replay()
            for k in range(args.steps % iters):
                step(*reqs[k])
            sync()
            t1 = time.perf_counter()
            g_elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(g_elapsed, op=dist.ReduceOp.MAX)
            # (fewer steps than one round: no replay took place, the steps ran eagerly over the direct exchange)
            mode, elapsed = ("hipgraph+direct-rccl" if args.steps >= iters else "eager+direct-rccl"), float(g_elapsed.item())
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] direct-RCCL graph path unavailable ({type(ex).__name__}: {ex}); reporting the eager path", file=sys.stderr)
            mod.direct = None
            torch.cuda.synchronize()
        finally:
            dog.cancel()

    # N > 1: the two exchanges of a step in isolation (xGMI all-to-all bandwidth vs the link roofline)
    a2a = None
    if sharded:
        try:
            a2a = {}
            for name, numel, dtype in (("indices_in", B_local * POOL * world, torch.int64),
                                       ("pooled_out", B_local * D * world, torch.float32)):
                src = torch.zeros(numel, dtype=dtype, device=dev)
                dst = torch.empty_like(src)
                for _ in range(5):
                    dist.all_to_all_single(dst, src)
                sync()
                t0 = time.perf_counter()
                for _ in range(50):
                    dist.all_to_all_single(dst, src)
                sync()
                dt = (time.perf_counter() - t0) / 50
                sent = src.element_size() * numel * (world - 1) // world  # bytes this rank puts on the links
                a2a[name] = {"bytes_per_rank": sent, "us": round(dt * 1e6, 1), "GB/s_per_rank": round(sent / dt / 1e9, 2),
                             "frac_of_xgmi": round(sent / dt / 1e9 / (153.0 * max(1, min(world - 1, 7))), 4)}
            if getattr(mod, "direct", None) is not None:  # the same two messages through the direct RCCL route
                for name, numel, dtype in (("indices_in_direct", B_local * POOL * world, torch.int64),
                                           ("pooled_out_direct", B_local * D * world, torch.float32)):
                    src = torch.zeros(numel, dtype=dtype, device=dev)
                    dst = torch.empty_like(src)
                    for _ in range(5):
                        mod.direct.all_to_all(dst, src)
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        mod.direct.all_to_all(dst, src)
                    sync()
                    dt = (time.perf_counter() - t0) / 50
                    sent = src.element_size() * numel * (world - 1) // world
                    a2a[name] = {"bytes_per_rank": sent, "us": round(dt * 1e6, 1), "GB/s_per_rank": round(sent / dt / 1e9, 2),
                                 "frac_of_xgmi": round(sent / dt / 1e9 / (153.0 * max(1, min(world - 1, 7))), 4)}
            a2a["note"] = ("eager all_to_all_single over RCCL, barrier-synchronised loop of 50; link roofline = 153 GB/s x "
                           "peers (MI355X_MICROARCH.md); messages this small are latency-bound")
        except Exception as ex:  # noqa: BLE001
            a2a = {"error": f"{type(ex).__name__}: {ex}"}

    # second, untimed pass: per-kernel breakdown (all kernel slots bracketed)
    E.profile_reset()
    E.profile_enable(0x3F)
    for k in range(min(args.steps, 50)):
        step(*reqs[k % iters])
    sync()
    E.profile_enable(0)
    names = ["fwd_contract", "bwd_contract", "reduce_apply", "plan", "bag_pool", "cache_gather"]
    breakdown = {}
    for w, nm in enumerate(names):
        n, ms = E.profile_read(w)
        if n:
            breakdown[nm + "_us"] = round(ms / n * 1e3, 2)

    if rank == 0:
        line = build_line(mode, elapsed, breakdown, a2a)
        if not sharded and not args.no_cpu_baseline and ntab == 1:
            line["cpu_baseline"] = cpu_baseline(reqs_np, cores_np, d_out_np)
        if args.run_baseline and not sharded and ntab == 1:
            line["dense_embedding_bag"] = dense_baseline(E_, D, reqs, grad, args.steps, args.warmup)
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        # (no destroy_process_group / communicator teardown: both were seen to hang on this stack; exit ends them)
        sys.stdout.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
