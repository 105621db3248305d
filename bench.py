#!/usr/bin/env python3
"""bench.py -- fwd+bwd throughput of the TT-EmbeddingBag hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward + backward (fused SGD) of the module over one batch of
synthetic lookups already resident in HBM, i.e. what the reference's own
benchmark times (tt_embeddings_benchmark.py:183-187).

default workload (cfg2)
  N = 1 : BASELINE.json configs[1], the repo benchmark config (E=11M, D=64, p=[200,220,250], q=[4,4,4],
          ranks=[32,32], B=512, L=20 -> nnz=10240, sparse SGD, use_cache=True but never populated -- exactly
          what the reference benchmark instantiates, :166-175 -- so every step also runs the hash-table
          frequency update).
  N > 1 : N such tables, one per rank (table-sharded, ttx_sharded.py), the 512-bag batch split across
          ranks, RCCL all-to-all of indices in / pooled vectors out.  Per-GPU lookups stay 10240 per step
          ("weak").
--workload cfg5 (BASELINE.json configs[4]): 26 tables of that shape, GLOBAL batch 4096, tables sharded
          t -> rank t % N (8 ranks: 4,4,3,3,3,3,3,3), each rank feeds 4096 / N bags per table; total work
          fixed ("strong").  N = 1 runs the same 26-table batch on one GPU.

value = true algorithmic GFLOP/s = 3 * 2*(q0 r1 q1 r2 + q0 q1 r2 q2) * nnz / time (the reference's formula,
:154-158/:190, WITHOUT its x iters slip; the README's 2657.6 "GFLOPS" is 265.8 on this scale -- BASELINE.md).
The timed region of K steps is repeated --repeats times; `value` / `ms_per_step` are the MEDIAN region,
`spread` holds min / max.  ONE JSON line by rank 0, with the roofline of the dominant kernel (backward
contraction, timed live with HIP events on its stream; `traffic` from the rocprofv3 PMC passes of
scripts/measure_traffic.sh if they were taken on THIS build) and a CPU baseline (oracle/ttx_cpu_baseline.c on
all host cores, bounded sample).
"""
import argparse
import glob
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "fbtt-embedding_amd"), os.path.join(ROOT, "tests"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Kernel arguments of eager launches in device memory (the HIP runtime reads this once, when it starts): on this stack the default
# puts them in host memory and every kernel's first s_load crosses the bus -- the free-running eager loop (the reference
# benchmark's form) measured 0.068 -> 0.052 ms/step at cfg2 (scripts/probes/r06_kernarg_ab.sh); a replayed hipGraph keeps its
# arguments on the device either way (0.0417 / 0.0416).  A process setting of the runtime, not a library behaviour: the module
# never touches the environment; the line reports what was in effect (`env`).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

P_SHAPES = [200, 220, 250]
POOL = 20
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 MFMA/VALU dense peak (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0        # HBM3E spec (6.3 TB/s achievable), same guide
XGMI_GBS = 153.0          # per link, per direction

# --workload: cfg2 is the bench line (BASELINE.json configs[1]); the others are the rest of
# SURVEY.md section 8(d)'s measurement list, for profiles/ -- never the driver's default.
WORKLOADS = {
    "cfg2": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "cfg3": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.2, populate=True),
    # SURVEY.md 8(d) names two exponents for cfg3's Zipf stream (the reference never sets one): 1.2 above, 1.05 here
    "cfg3a105": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.05, populate=True),
    # cfg3's index stream before the cache is populated: every hot row goes through the contraction
    "cfg3warm": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.2, populate=False),
    # ... with duplicate lookups sharing their contraction (TTEmbeddingBag(dedup=True))
    "cfg3warm-dedup": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.2, populate=False, dedup=True),
    "cfg2-dedup": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False, dedup=True),
    "cfg4": dict(q=[4, 4, 8], ranks=[64, 64], tables=1, B=512, optimizer="adagrad", alpha=1.0, populate=False),
    # BASELINE.json configs[4]: 26 tables, global batch 4096; shards over the ranks it is launched with
    "cfg5": dict(q=[4, 4, 4], ranks=[32, 32], tables=26, B=4096, optimizer="sgd", alpha=1.0, populate=False),
    "cfg5full": dict(q=[4, 4, 4], ranks=[32, 32], tables=26, B=4096, optimizer="sgd", alpha=1.0, populate=False),
    # one rank's share of cfg5 at 8 GPUs: 4 of the 26 tables, the whole 4096-bag batch
    "cfg5shard": dict(q=[4, 4, 4], ranks=[32, 32], tables=4, B=4096, optimizer="sgd", alpha=1.0, populate=False),
    # ONE table of that share (81,920 lookups): what a launch set per table group would run (why the
    # pooled exchange of one group is not hidden under the lookup of the next)
    "cfg5shard1": dict(q=[4, 4, 4], ranks=[32, 32], tables=1, B=4096, optimizer="sgd", alpha=1.0, populate=False),
    # shapes outside the specialised family (reference-default q for D = 32 has q0 = 2): generic kernels
    "d32": dict(q=[2, 4, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d16": dict(q=[2, 2, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d256": dict(q=[4, 8, 8], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d32q4": dict(q=[4, 2, 4], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d128r32": dict(q=[4, 4, 8], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # duplicate sharing where it can pay: a cfg5-size batch of a skewed stream (26 tables x 4096 bags x 20 lookups, Zipf 1.2)
    "cfg5z": dict(q=[4, 4, 4], ranks=[32, 32], tables=26, B=4096, optimizer="sgd", alpha=1.2, populate=False),
    "cfg5z-dedup": dict(q=[4, 4, 4], ranks=[32, 32], tables=26, B=4096, optimizer="sgd", alpha=1.2, populate=False, dedup=True),
    "tb4z": dict(q=[4, 4, 4], ranks=[32, 32], tables=4, B=4096, optimizer="sgd", alpha=1.2, populate=False),
    "tb4z-dedup": dict(q=[4, 4, 4], ranks=[32, 32], tables=4, B=4096, optimizer="sgd", alpha=1.2, populate=False, dedup=True),
    # generic kernels: ranks beyond the LDS (core 1 walked in K blocks x column passes), and the reference tests' odd ranks
    "r128": dict(q=[4, 4, 4], ranks=[128, 128], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "r13": dict(q=[4, 4, 4], ranks=[13, 12], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d512": dict(q=[8, 8, 8], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # the reference's default factorings of D = 768 / 1024: q2 = 12 / 16 (round 4: templates with q2 <= 16 at ranks <= 32)
    "d768": dict(q=[8, 8, 12], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d1024": dict(q=[8, 8, 16], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # ... of D = 320 / 448: q0 = 5 / 7 has no exact part split -- core 0 zero-padded to 8 slots (round 4)
    "d320": dict(q=[5, 8, 8], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d448": dict(q=[7, 8, 8], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d1024r64": dict(q=[8, 8, 16], ranks=[64, 64], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # default factorings that were on the generic kernels until round 5: a prime last factor (q2 = 32 templates), q1 = 9 / 10 (q1 = 16
    # templates); d368 -- q1 = 16 AND a prime last factor -- still is
    "d272": dict(q=[4, 4, 17], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d368": dict(q=[1, 16, 23], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d720": dict(q=[8, 9, 10], ranks=[32, 32], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "d256b1024": dict(q=[4, 8, 8], ranks=[32, 32], tables=1, B=1024, optimizer="sgd", alpha=1.0, populate=False),
    "r96": dict(q=[4, 4, 4], ranks=[96, 96], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "r256": dict(q=[4, 4, 4], ranks=[256, 256], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "tb16": dict(q=[4, 4, 4], ranks=[32, 32], tables=16, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "tb8": dict(q=[4, 4, 4], ranks=[32, 32], tables=8, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "tb4": dict(q=[4, 4, 4], ranks=[32, 32], tables=4, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # two and four cores (the reference contracts 2, 3 and 4: tt_embeddings_cuda.cu:754-776, tt_embeddings_test.py:65-70), the same
    # 11M-row table: p = [3317, 3317] / [58, 58, 58, 58]
    "t2": dict(q=[8, 8], ranks=[32], p=[3317, 3317], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # ... the default four-core factoring of D = 256 (merged last factor q2 q3 = 16)
    "t4d256": dict(q=[4, 4, 4, 4], ranks=[32, 32, 32], p=[58, 58, 58, 58], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "t4": dict(q=[2, 4, 4, 2], ranks=[32, 32, 32], p=[58, 58, 58, 58], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    # ... of D = 512 (merged last factor 32: round 5)
    "t4d512": dict(q=[4, 4, 4, 8], ranks=[32, 32, 32], p=[58, 58, 58, 58], tables=1, B=512, optimizer="sgd", alpha=1.0, populate=False),
    "t2big": dict(q=[8, 8], ranks=[32], p=[3317, 3317], tables=4, B=4096, optimizer="sgd", alpha=1.0, populate=False),
    "t4big": dict(q=[2, 4, 4, 2], ranks=[32, 32, 32], p=[58, 58, 58, 58], tables=4, B=4096, optimizer="sgd", alpha=1.0, populate=False),
}


def flop_per_nnz_fwd(q, r):
    """the reference benchmark's count (tt_embeddings_benchmark.py:154-158: q0 r1 q1 r2 + q0 q1 r2 q2 multiply-adds for three
    cores), stated for any number of cores: contraction step t multiplies [q_0 .. q_{t-1} x r_t] by core t's [r_t x q_t r_{t+1}]"""
    rr = list(r) + [1]
    return 2.0 * sum(float(np.prod(q[:t])) * rr[t - 1] * q[t] * rr[t] for t in range(1, len(q)))


def flop_per_nnz_fwd_executed(q, r):
    """what the kernels here execute per lookup (forward).  Two and three cores: the reference's left-to-right count.  Four cores with
    q2 q3 <= 16 (the route through the three-core kernels): the last two cores are contracted first --
    r2 q2 r3 q3 multiply-adds for M = core_2 . core_3 -- and the lookup is a three-core lookup with the merged last factor."""
    if len(q) == 4:
        # (round 6, advisor: the predicate is the LIBRARY's -- csrc/ttx_tt.hip t4_merge_dims, q2 q3 <= 32 with q3 <= 8 where a
        #  three-core template holds the merged shape -- asked through its stateless tile query: no generic walk = the merged route)
        import tt_embeddings as E

        if E.debug_tiles(1, [4, 4, 4, 4], list(q), [1] + list(r) + [1])["MC"] == 0:
            return 2.0 * (r[1] * q[2] * r[2] * q[3]) + flop_per_nnz_fwd([q[0], q[1], q[2] * q[3]], [r[0], r[1]])
    return flop_per_nnz_fwd(q, r)


def source_hash():
    """identifies the kernel build a PMC measurement belongs to (same on the build container and the GPU box)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "fbtt-embedding_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def dense_baseline(E_, D, reqs, grad, steps, warmup):
    """the table the TT cores replace: nn.EmbeddingBag(E, D, mode="sum", sparse=True) + SGD(lr=0.1) on the same
    requests, eager, as tt_embeddings_benchmark.py:195-211 does behind --run-baseline"""
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


def cpu_baseline(requests, cores, d_out, q, ranks, B, budget_s=12.0):
    """oracle/ttx_cpu_baseline.c -- the CPU restatement, OpenMP over the lookups on all host cores -- on the same
    requests: forward + fused-SGD backward (the sequential parity oracle, 1 thread, is timed beside it)"""
    import oracle_lib as O

    g = O.make_geom(1, P_SHAPES, q, ranks)
    D = int(np.prod(q))
    fl = 3.0 * flop_per_nnz_fwd(q, ranks)
    cores = [c.copy() for c in cores]
    step = O.OmpStep(g, B, D, max(i.size for i, _ in requests), cores)
    rows = [O.rowidx_from_offsets(off, 1) for _, off in requests]
    step(O.OPTIM_SGD, 0.1, 0.0, requests[0][0], requests[0][1], rows[0][0], rows[0][1], d_out, cores)  # warm-up (thread pool)
    done, nnz, t0 = 0, 0, time.perf_counter()
    while True:
        idx, off = requests[done % len(requests)]
        step(O.OPTIM_SGD, 0.1, 0.0, idx, off, rows[done % len(requests)][0], rows[done % len(requests)][1], d_out, cores)
        done += 1
        nnz += idx.size
        el = time.perf_counter() - t0
        if el > budget_s or done >= 2000:
            break
    # the scalar parity oracle on one core, a couple of steps, for scale
    c1 = [c.copy() for c in cores]
    s0 = time.perf_counter()
    for k in range(3):
        idx, off = requests[k % len(requests)]
        O.tt_forward(g, B, D, idx, rows[k % len(requests)][0], rows[k % len(requests)][1], c1)
        O.tt_backward(g, O.OPTIM_SGD, B, D, 0.1, 0.0, idx, rows[k % len(requests)][0], rows[k % len(requests)][1], d_out, c1)
    one = fl * 3 * requests[0][0].size / (time.perf_counter() - s0) / 1e9
    return {"value": round(fl * nnz / el / 1e9, 3), "unit": "GFLOP/s", "cores": step.threads, "kind": "port",
            "sample": f"{done} fwd+bwd(SGD) steps of the same requests ({nnz} lookups) in {el:.1f} s, oracle/ttx_cpu_baseline.c "
                      f"(OpenMP over lookups, {step.threads} threads of {os.cpu_count()} host cpus); {el / nnz * 1e6:.3f} us/nnz",
            "single_thread_oracle_gflops": round(one, 3)}


def secondary_record(workload="cfg5shard", steps=40, repeats=3):
    """The regime where the contraction kernels -- not the launches -- are the bound, beside the default line: one rank's share of
    BASELINE.json configs[4] at 8 GPUs (4 of the 26 tables, the whole 4096-bag batch: 327,680 lookups per step), the same
    bench.py in a process of its own (a second module of that size beside the first would only perturb both).  Returns the
    fields of that line the driver's file should carry: ms_per_step / value (hipGraph replay), the live HIP-event roofline of
    the backward contraction, per-kernel event brackets."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", "10", "--repeats",
           str(repeats), "--no-cpu-baseline", "--no-secondary"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        keep = ("metric", "value", "unit", "ms_per_step", "steps", "repeats", "timed_mode", "eager_ms_per_step", "us_per_nnz", "kernel_us",
                "roofline", "spread", "no_prefetch", "cache_gather_roofline", "cache_hit_rate", "dtype", "kernel_rooflines")
        rec = {k: j[k] for k in keep if k in j}
        rec["workload"] = j["config"]["workload"]
        fl = j["config"]["flop_per_nnz_fwd_bwd"] * j["config"]["nnz_per_step_total"]
        rec["step_frac_of_fp32_peak"] = round(fl / (j["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)
        if "fwd_contract_us" in j.get("kernel_us", {}):
            rec["fwd_frac_of_fp32_peak"] = round(fl / 3.0 / (j["kernel_us"]["fwd_contract_us"] * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4)
        if workload == "cfg5shard":
            rec["predicted_cfg5_on_8_gpus"] = predict_cfg5_8gpu(j["ms_per_step"])
        return rec
    except Exception as ex:  # noqa: BLE001 -- the secondary record must never take the headline line with it
        return {"error": f"{type(ex).__name__}: {ex}", "workload": workload}


def predict_cfg5_8gpu(shard_ms, exchange_latency_us=20.0):
    """A WRITTEN PREDICTION of BASELINE.json configs[4] on 8 x MI355X (no multi-GPU run has happened in any round: the first real
    one has this to be compared with).  26 tables on 8 ranks = 4,4,3,3,3,3,3,3 per owner; the step is as long as
    the slowest owner's: a 4-table owner's local step -- exactly the cfg5shard workload measured in this job, `shard_ms` -- plus
    the three exchanges of a step (indices in: int32 since round 6; pooled rows out; their gradient back), each one a group of 7
    concurrent point-to-point transfers over 7 links of 153 GB/s, priced at latency + bytes over the slowest link."""
    W, NT, Bg, D, L = 8, 26, 4096, 64, POOL
    b_local = Bg // W
    own = 4                                                           # tables of the busiest owner
    idx_link = NT * b_local * L * 4 / W                               # bytes a rank sends to ONE peer (its batch for that peer's ~NT/W tables)
    pooled_link = own * b_local * D * 4                               # bytes the busiest owner returns to ONE peer
    ex_us = [exchange_latency_us + idx_link / (XGMI_GBS * 1e3), exchange_latency_us + pooled_link / (XGMI_GBS * 1e3),
             exchange_latency_us + pooled_link / (XGMI_GBS * 1e3)]
    step_ms = shard_ms + sum(ex_us) * 1e-3
    lookups = NT * Bg * L
    fl = 3.0 * flop_per_nnz_fwd([4, 4, 4], [32, 32]) * lookups
    return {"ms_per_step": round(step_ms, 4), "value_gflops": round(fl / (step_ms * 1e-3) / 1e9, 1),
            "exchange_us": [round(x, 1) for x in ex_us], "bytes_per_link": {"indices_int32": int(idx_link), "pooled_rows": int(pooled_link)},
            "assumes": f"{exchange_latency_us} us launch + rendezvous latency per exchange (one-rank RCCL calls measure ~5 us of issue "
                       "time; the rest is a guess until two devices have talked), transfers at the 153 GB/s link rate, no overlap "
                       "of an exchange with the local lookup, the busiest owner (4 of 26 tables) sets the pace",
            "ideal_speedup_over_one_gpu": round(NT / own, 2)}


def self_launch(n):
    """`python bench.py --gpus N` started by itself (no WORLD_SIZE in the environment): one rank per GPU under
    torch.distributed.run on 127.0.0.1 with the same arguments -- the command the driver uses for --gpus 1 works unchanged for
    --gpus 2/4/8.  Returns the exit code of the job; rank 0's JSON line goes to stdout as it is."""
    import socket
    import subprocess

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f"[bench] --gpus {n} needs {n} visible GPUs; this box shows {have} "
              f"(torch.cuda.device_count(); HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')})",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as so:  # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def n1_record(workload, steps, warmup, device_index):
    """The SAME workload on ONE GPU (rank 0's device, after the other ranks have left it alone): the denominator of the scaling
    efficiency, measured in the same job on the same box.  A process of its own, like secondary_record."""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "TTX_FORCE_EXCHANGE")}
    vis = os.environ.get("HIP_VISIBLE_DEVICES")
    env["HIP_VISIBLE_DEVICES"] = vis.split(",")[device_index] if vis else str(device_index)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
           "--repeats", "3", "--no-cpu-baseline", "--no-secondary"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "timed_mode": j["timed_mode"],
                "workload": j["config"]["workload"], "nnz_per_step_total": j["config"]["nnz_per_step_total"],
                "what": "bench.py --gpus 1 of the same workload, run by rank 0 on its own GPU after the sharded region (same box, same job)"}
    except Exception as ex:  # noqa: BLE001 -- never take the sharded line with it
        return {"error": f"{type(ex).__name__}: {ex}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--no-cache", action="store_true", help="use_cache=False (skip the hash-table frequency update)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time the eager Python path only (no hipGraph replay)")
    ap.add_argument("--prefetch", default="round", choices=["round", "next", "none"],
                    help="captured round: lookup prologues of the round's batches in one launch up front (round, default), "
                         "of the next batch on a side stream under the current backward (next), or in line (none)")
    ap.add_argument("--optimizer", default=None, choices=["sgd", "adagrad"])
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--run-baseline", action="store_true",
                    help="also time the uncompressed nn.EmbeddingBag(E, D, sparse) + SGD on the same requests "
                         "(tt_embeddings_benchmark.py --run-baseline; needs E*D*4 bytes per table)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default workload only: skip the appended large-batch record (cfg5's per-GPU shard, 327,680 lookups per step)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="(test) take the N > 1 code path -- process group, sharded module, all-to-all -- with one rank")
    ap.add_argument("--no-n1", action="store_true",
                    help="N > 1: skip rank 0's one-GPU run of the same workload (the `n1` record of the line)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by itself: spawn the ranks (the driver's `python bench.py --gpus N` and its torch.distributed.run form both work)
        sys.exit(self_launch(args.gpus))
    wl = WORKLOADS[args.workload]
    global P_SHAPES
    P_SHAPES = wl.get("p", P_SHAPES)
    Q_SHAPES, RANKS, B_GLOBAL = wl["q"], wl["ranks"], wl["B"]
    if args.optimizer is None:
        args.optimizer = wl["optimizer"]
    ntab = wl["tables"]
    if args.gpus > 1 and args.workload not in ("cfg2", "cfg5"):
        raise SystemExit("--gpus N > 1 runs the cfg2-per-rank (default) or the cfg5 sharded workload only")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = world > 1 or args.force_sharded
    if args.force_sharded:
        os.environ["TTX_FORCE_EXCHANGE"] = "1"
        os.environ.setdefault("MASTER_PORT", "29561")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (args.force_sharded and args.gpus == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` (it spawns its ranks) or "
                         f"torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL brings up 128 channels per communicator by default on this GPU; in the sandboxed boxes of this
        # pool that alone took minutes (measured: > 120 s at world size 1, 3.6 s with 4 channels).  The two
        # exchanges of a step are < 1 MB per peer: a handful of channels carries them.
        # That cap is a bring-up-time fix measured on ONE rank; with real peers the channel count is RCCL's to choose
        # (8 ranks x 7 xGMI links: nobody measured 8 channels there), so it is only applied to the one-rank run.
        if world == 1:
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        dist.init_process_group("nccl", device_id=dev)

    import gen_inputs as G
    import tt_embeddings as E
    import tt_embeddings_ops as ops
    import ttx_sharded

    # the library's process-global test / ablation knobs (ttx_debug_skip, ttx_set_chunk, ...) must be at their defaults: a line
    # timed with phases skipped or tiles overridden is not a measurement.  Ablation scripts say TTX_ALLOW_DEBUG=1 and get a line
    # that carries the mask and the word INVALID.
    debug_state = int(E.lib().ttx_debug_state())
    if debug_state and not os.environ.get("TTX_ALLOW_DEBUG"):
        raise SystemExit(f"bench.py: libttx's debug knobs are set (ttx_debug_state() = {debug_state}; TTX_DEBUG_SKIP / TTX_LDS_BUDGET in "
                         f"the environment?) -- refusing to time an ablated build; TTX_ALLOW_DEBUG=1 overrides (the line is marked INVALID)")

    E_, D = int(np.prod(P_SHAPES)), int(np.prod(Q_SHAPES))
    opt = ops.OptimType.SGD if args.optimizer == "sgd" else ops.OptimType.EXACT_ADAGRAD
    iters = 10  # request batches, like the reference's --iters
    hit_rate = None
    n_tt_per_step = None
    cfg5 = args.workload == "cfg5"
    assert B_GLOBAL % world == 0, "the global batch must split evenly over the ranks"
    B_local = B_GLOBAL // world
    torch.manual_seed(1234 + rank)
    reqs_np = cores_np = d_out_np = None
    if not sharded:
        use_cache = (not args.no_cache) and ntab == 1
        kw = dict(sparse=True, optimizer=opt, learning_rate=0.1, use_cache=use_cache, weight_dist="uniform", device=dev,
                  dedup=bool(wl.get("dedup", False)))
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
        tables_total, owned = ntab, [ntab]
        if wl["populate"]:
            # warm the frequency table on a DIFFERENT request stream (same distribution), then populate:
            # the timed batches then mix cache hits (hot rows) with TT lookups (the Zipf tail)
            warm = G.make_requests(4321, 5 * iters, B_GLOBAL, ntab, POOL, E_, alpha=wl["alpha"])
            for i, o in warm:
                step(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev))
            mod.cache_populate()
            n_tt = sum(E.preprocess_indices_sync(i, o, 1, False, mod.hashtbl, mod.cache_state)[3] for i, o in reqs)
            hit_rate = 1.0 - n_tt / float(iters * nnz_step_total)
            n_tt_per_step = n_tt / float(iters)  # lookups the contraction kernels actually see per launch (the misses)
    else:
        # cfg2 (default): `world` tables, one per rank.  cfg5: 26 tables, t -> rank t % world (uneven ownership).
        tables_total = ntab if cfg5 else world
        mod = ttx_sharded.ShardedTableBatchedTTEmbeddingBag(
            tables_total, E_, D, RANKS, tt_p_shapes=P_SHAPES, tt_q_shapes=Q_SHAPES, sparse=True, optimizer=opt,
            learning_rate=0.1, use_cache=False, weight_dist="uniform", device=dev)
        owned = [len(o) for o in mod.owned]
        reqs_np = G.make_requests(1235 + rank, iters, B_local, tables_total, POOL, E_)
        reqs = [(torch.from_numpy(i).to(dev), torch.from_numpy(o).to(dev)) for i, o in reqs_np]
        grad = torch.from_numpy(G.make_grad(1236 + rank, tables_total, B_local, D)).to(dev)
        step = lambda i, o: mod(i, o, fixed_pooling=POOL).backward(grad)  # noqa: E731
        nnz_step_total = tables_total * B_GLOBAL * POOL  # every table sees the whole global batch

    rank_regions = []  # sharded: per timed region, every rank's elapsed seconds
    reported_ranks = []  # ... of the last region of the REPORTED mode

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(run, n):
        """the contract's bracket: barrier + synchronize on both sides, MAX over the ranks"""
        sync()
        t0 = time.perf_counter()
        run(n)
        sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if sharded:
            every = [torch.zeros_like(el) for _ in range(world)]
            dist.all_gather(every, el)  # every rank's own clock around the region (the line carries them beside the MAX)
            rank_regions.append([float(e.item()) for e in every])
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def eager_steps(n):
        for k in range(n):
            step(*reqs[k % iters])

    eager_steps(args.warmup)
    sync()

    # ---- region 1 (eager Python path): live HIP-event timing of the dominant kernel ----
    E.profile_reset()
    E.profile_enable(1 << E.PROF_BWD)
    eager_elapsed = timed(eager_steps, args.steps)
    E.profile_enable(0)
    n_bwd, ms_bwd = E.profile_read(E.PROF_BWD)
    bwd_src = "HIP events around each launch of the eager region"
    # ... and the same loop with no event brackets in the stream: THE eager figure (`eager_ms_per_step`).  A timing event pair per
    # launch costs the stream ~2 x 10 us of marker packets -- the bracketed region above ran at about half this one's rate and was
    # what rounds 1-4 (and this round's first lines) reported as the eager step.
    eager_profiled = eager_elapsed
    eager_elapsed = timed(eager_steps, args.steps)
    eager_regions = [eager_elapsed]
    # ... and once more with the OPT-IN direct backward (round 6: importing the module no longer wraps torch.Tensor.backward;
    # ops.enable_direct_backward() does, explicitly): `out.backward(grad)` of the lookup's own output calls its node on the
    # calling thread instead of going through autograd's engine.  Reported beside the plain figure, never instead of it.
    eager_direct = None
    if not sharded and hasattr(ops, "enable_direct_backward"):
        ops.enable_direct_backward()
        try:
            eager_steps(min(args.warmup, 10))
            sync()
            eager_direct = timed(eager_steps, args.steps)
        finally:
            ops.disable_direct_backward()

    def build_line(mode, regions, breakdown, a2a, note=None):
        fl_fwd = flop_per_nnz_fwd(Q_SHAPES, RANKS)
        elapsed = statistics.median(regions)
        ms_per_step = elapsed / args.steps * 1e3
        gflops = 3.0 * fl_fwd * nnz_step_total / (elapsed / args.steps) / 1e9
        # lookups the dominant kernel of THIS rank contracts per launch (rank 0 owns the most tables)
        rank0_nnz = (owned[0] if sharded else tables_total) * B_GLOBAL * POOL
        # a KERNEL's roofline counts the lookups the launch contracts: with a live cache only the misses reach the contraction
        # (cfg3: ~12 % of the batch) -- the step-level GFLOP/s above keeps the full numerator, this does not
        launch_nnz = n_tt_per_step if n_tt_per_step is not None else rank0_nnz
        bwd_flop_per_launch = 2.0 * fl_fwd * launch_nnz  # backward contraction = 2/3 of the algorithmic fwd+bwd FLOP
        bwd_us = ms_bwd / max(n_bwd, 1) * 1e3
        achieved = bwd_flop_per_launch / (bwd_us * 1e-6) / 1e12 if n_bwd else 0.0
        traffic, traffic_note = None, "no PMC measurement (scripts/measure_traffic.sh) for this workload"
        pmc = os.path.join(ROOT, "profiles", "pmc_bwd_bytes.json")
        if os.path.exists(pmc) and args.workload == "cfg2" and not sharded:
            try:
                j = json.load(open(pmc))
                if j.get("source_hash") == source_hash():
                    traffic, traffic_note = j.get("hbm_bytes_per_launch"), j.get("source", "")
                else:
                    traffic_note = "profiles/pmc_bwd_bytes.json was measured on another build of the kernels (source hash differs): not reported"
            except Exception:  # noqa: BLE001
                pass
        pmc_all = os.path.join(ROOT, "profiles", "pmc_bytes.json")  # every measured workload (scripts/measure_traffic.sh --workload W)
        rocprof_us, rocprof_note = None, "no rocprofv3 summary of this build under profiles/ (scripts/regen_profiles.sh)"
        if os.path.exists(pmc_all) and not sharded:
            try:
                j = json.load(open(pmc_all))
                if j.get("source_hash") == source_hash():
                    w_ = j["workloads"].get(args.workload, {})
                    kk = next((k for k in w_.get("kernels", {}) if k.startswith(("spec_bwd_kernel", "bwd_kernel", "t2_bwd_kernel"))), None)
                    if kk:
                        rec_ = w_["kernels"][kk]
                        traffic = rec_["hbm_bytes_per_launch"]
                        traffic_note = (f"{w_.get('source', '')}; {kk}: FETCH_SIZE x2 + WRITE_SIZE, per-launch average over "
                                        f"{rec_.get('pmc_launches')} launches (profiles/pmc_bytes.json)")
                        if rec_.get("rocprof_avg_us") and len(Q_SHAPES) != 4:
                            rocprof_us = float(rec_["rocprof_avg_us"])
                            rocprof_note = f"rocprofv3 --kernel-trace --stats, {rec_.get('rocprof_calls')} launches ({w_.get('source', '')})"
                elif traffic is None:
                    traffic_note = "profiles/pmc_bytes.json was measured on another build of the kernels (source hash differs): not reported"
            except Exception:  # noqa: BLE001
                pass
        if rocprof_us is None and os.path.exists(pmc) and args.workload == "cfg2" and not sharded:
            try:
                j = json.load(open(pmc))
                if j.get("source_hash") == source_hash() and j.get("rocprof_avg_us"):
                    rocprof_us, rocprof_note = float(j["rocprof_avg_us"]), j.get("rocprof_source", "")
                elif j.get("source_hash") != source_hash():
                    rocprof_note = "profiles/pmc_bwd_bytes.json belongs to another build of the kernels (source hash differs)"
            except Exception:  # noqa: BLE001
                pass
        rk = os.path.join(ROOT, "profiles", "rocprof_kernels.json")
        if rocprof_us is None and os.path.exists(rk) and args.workload != "cfg2" and not sharded and len(Q_SHAPES) != 4:
            # the other workloads: scripts/kprof.sh's per-kernel durations of THIS build (eager launches); the backward contraction is
            # one launch per step there (four cores: its FLOP are spread over helper kernels -- no single-kernel figure)
            try:
                j = json.load(open(rk))
                if j.get("source_hash") == source_hash():
                    ks = j["workloads"].get(args.workload, {})
                    us = next((v["avg_us"] for k, v in ks.items() if k.startswith(("spec_bwd_kernel", "bwd_kernel", "t2_bwd_kernel"))), None)
                    if us:
                        rocprof_us, rocprof_note = float(us), f"rocprofv3 --kernel-trace --stats, {j.get('how', 'scripts/kprof.sh')} (profiles/rocprof_kernels.json)"
                else:
                    rocprof_note = "profiles/rocprof_kernels.json belongs to another build of the kernels (source hash differs)"
            except Exception:  # noqa: BLE001
                pass
        if sharded:
            own_txt = f"; {tables_total} tables sharded t -> rank t % {world} (tables per rank {owned}), RCCL all-to-all; B_local={B_local}"
        else:
            own_txt = ""
        cache_txt = ("False" if (args.no_cache or sharded or ntab > 1) else
                     (f"True(populated from 50 other batches, 256Ki rows, Zipf a={wl['alpha']}, hit rate {hit_rate:.3f})" if wl["populate"]
                      else "True(unpopulated)"))
        what = "TTEmbeddingBag" if tables_total == 1 else f"TableBatchedTTEmbeddingBag x{tables_total} tables"
        graph_txt = ""
        if mode.startswith("hipgraph"):
            graph_txt = "; timed as hipGraph replay of the captured module fwd+bwd steps (one graph per round of the 10 request batches)"
            if "rccl" in mode:
                graph_txt += "; all-to-all exchanges issued on RCCL directly, inside the graph"
            if "prefetch-round" in mode:
                graph_txt += ("; the lookup prologues (frequency update, bag rows, plan) of the round's 10 batches are enqueued "
                              "up front in one launch (module.prefetch_many), every replay plans them again")
                if "rccl" in mode:
                    graph_txt += "; the round's index exchange too: one all-to-all for the 10 batches"
            elif "prefetch-next" in mode:
                graph_txt += ("; the lookup prologue of batch k+1 runs on a side stream under the backward of batch k "
                              "(module.prefetch), captured as a forked branch")
        line = {
            "metric": ("fwd+bwd GFLOPS (true algorithmic: 3 x fwd FLOP / time), TT-EmbeddingBag E=11M "
                       f"D={D} ranks={RANKS} nnz={nnz_step_total // world if not cfg5 else nnz_step_total}"),
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if cfg5 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.workload}: {what} E={E_} D={D} p={P_SHAPES} q={Q_SHAPES} ranks={RANKS} "
                                    f"B={B_GLOBAL} L=20 nnz/step={nnz_step_total} sparse {args.optimizer.upper()}, use_cache={cache_txt}{own_txt}"),
                       "nnz_per_step_total": nnz_step_total, "flop_per_nnz_fwd_bwd": 3.0 * fl_fwd,
                       # `value` counts the REFERENCE's left-to-right FLOP (its benchmark's formula) whatever order the kernels
                       # contract in: for four cores that is reference-equivalent throughput, not matrix-pipe utilisation
                       "flop_per_nnz_fwd_reference": fl_fwd, "flop_per_nnz_fwd_executed": flop_per_nnz_fwd_executed(Q_SHAPES, RANKS),
                       **({"flop_note": "four cores: value is reference-equivalent GFLOP/s (left-to-right count); the kernels contract "
                                        "the last two cores first and execute flop_per_nnz_fwd_executed per lookup -- no fraction "
                                        "of the fp32 peak is claimed for this line"} if len(Q_SHAPES) == 4 else {}),
                       "path": "Python module -> C++ autograd node (or ctypes) -> C ABI -> HIP" + (graph_txt or "; eager")},
            "timed_mode": mode,
            "repeats": len(regions),
            "spread": {"ms_per_step_min": round(min(regions) / args.steps * 1e3, 4), "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
                       "region_ms": [round(r * 1e3, 3) for r in regions]},
            "no_prefetch": (None if not plain_regions else
                            {"ms_per_step": round(statistics.median(plain_regions) / args.steps * 1e3, 4),
                             "value": round(3.0 * fl_fwd * nnz_step_total / (statistics.median(plain_regions) / args.steps) / 1e9, 2),
                             "what": "the same captured round with every step's prologue in line (no side stream)"}),
            "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4),
            "eager_value": round(3.0 * fl_fwd * nnz_step_total / (eager_elapsed / args.steps) / 1e9, 2),
            "eager_what": ("the reference benchmark's loop form, `tt_emb(indices, offsets).backward(grad)` per request "
                           "(tt_embeddings_benchmark.py:94-108), free-running, no graph, no planning ahead, no event brackets, "
                           "backward() through autograd's engine (the module as imported)"),
            "env": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
            "eager_with_event_brackets_ms_per_step": round(eager_profiled / args.steps * 1e3, 4),
            "eager_direct_backward": (None if eager_direct is None else {
                "ms_per_step": round(eager_direct / args.steps * 1e3, 4),
                "what": "the same loop after ops.enable_direct_backward() (opt-in, off by default: wraps torch.Tensor.backward so "
                        "that backward() of a fused-optimizer lookup's own output skips autograd's engine)"}),
            "us_per_nnz": round(elapsed / args.steps / nnz_step_total * 1e6, 5),
            "ref_formula_gflops_x_iters": round(gflops * 10, 1),
            "reference_readme_true_gflops": 265.8,
            "kernel_us": breakdown,
            # ONE figure per kernel (round 5): `achieved` / `frac` divide the algorithmic FLOP of a launch by the kernel's average
            # duration as rocprofv3 --kernel-trace --stats reports it for THIS build (profiles/, reproducible by hand from
            # <tag>_kernel_stats*.md); the live HIP-event bracket of this run -- which includes event overhead and, in an eager
            # region, launch gap -- stays beside it as `events_*` / `frac_events` and is what `frac` falls back to (and says so in
            # `frac_source`) when profiles/ holds no rocprofv3 summary of this build of the kernels.
            "roofline": (lambda ach_r: {
                "bound": "mfma", "kernel": "spec_bwd_kernel / bwd_kernel (backward contraction)",
                "achieved": round(ach_r if ach_r else achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round((ach_r if ach_r else achieved) / PEAK_FP32_TFLOPS, 5),
                "frac_source": ("rocprofv3 --kernel-trace --stats average of this build (profiles/)" if ach_r else
                                "live HIP events of this run (no rocprofv3 summary of this build under profiles/)"),
                "avg_us": rocprof_us if ach_r else round(bwd_us, 2), "rocprof_avg_us": rocprof_us, "rocprof_source": rocprof_note,
                "frac_events": round(achieved / PEAK_FP32_TFLOPS, 5), "events_avg_us": round(bwd_us, 2), "events_launches": n_bwd,
                "events_timed_by": bwd_src,
                "traffic": traffic, "traffic_source": traffic_note, "launches": n_bwd,
                "flop_per_launch": bwd_flop_per_launch, "lookups_per_launch": round(launch_nnz, 1),
                "kernel_build": source_hash()})(
                    (bwd_flop_per_launch / (rocprof_us * 1e-6) / 1e12) if rocprof_us else None),
            "kernel_us_note": ("HIP-event brackets around each launch in an eager pass: every bracket includes ~2 us of launch "
                               "gap, so their sum exceeds the replayed step; rocprofv3 durations: profiles/"),
        }
        if len(Q_SHAPES) == 2 and n_bwd:
            # two cores: a lookup is one [q0 x r1] x [r1 x q1] product -- 2 FLOP per byte fetched: the bound is bandwidth, not the
            # matrix pipe.  Algorithmic bytes of the backward per lookup: core-0 slice read + its partial row written (q0 r1 floats
            # each) + the bag's gradient row read (D floats); per chunk the core-1 slice read and its partial written.
            per = 4.0 * (2 * Q_SHAPES[0] * RANKS[0] + D)
            byt = per * rank0_nnz + 2 * 4.0 * RANKS[0] * Q_SHAPES[1] * (rank0_nnz / 16.0)
            gbs = byt / (bwd_us * 1e-6) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "t2_bwd_kernel (two-core backward)", "achieved": round(gbs, 1),
                                "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": round(gbs / (PEAK_HBM_TBS * 1e3), 5), "traffic": None,
                                "launches": n_bwd, "avg_us": round(bwd_us, 2), "timed_by": bwd_src, "bytes_per_launch": byt,
                                "note": "core slices of 1 KB per lookup come from L2 / Infinity Cache, the partial rows go to HBM",
                                "kernel_build": source_hash()}
        # (round 6) every kernel of the step against ITS bound, from the rocprofv3 durations and PMC bytes of this build
        # (profiles/pmc_bytes.json, scripts/measure_traffic.sh; absent or from another build: no entry): the contraction kernels
        # against the fp32 MFMA peak on their algorithmic FLOP, everything else against 8 TB/s on the HBM bytes the counters saw
        if os.path.exists(pmc_all) and not sharded and len(Q_SHAPES) == 3:
            try:
                j = json.load(open(pmc_all))
                if j.get("source_hash") == source_hash():
                    rows_ = []
                    for kname, rec_ in j["workloads"].get(args.workload, {}).get("kernels", {}).items():
                        us = rec_.get("rocprof_avg_us")
                        if not us or int(rec_.get("rocprof_calls") or 0) < 10:  # (one-off launches -- cache_populate's -- are not the step's)
                            continue
                        ent = {"kernel": kname[:60], "avg_us": us, "hbm_bytes": rec_.get("hbm_bytes_per_launch")}
                        if rec_.get("hbm_bytes_per_launch"):
                            ent["hbm_GBps"] = round(rec_["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9, 1)
                            ent["hbm_frac"] = round(ent["hbm_GBps"] / (PEAK_HBM_TBS * 1e3), 4)
                        if kname.startswith(("spec_fwd_kernel", "fwd_kernel")):
                            ent["mfma_frac"] = round(bwd_flop_per_launch / 2.0 / (us * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4)
                            ent["bound"] = "mfma"
                        elif kname.startswith(("spec_bwd_kernel", "bwd_kernel")):
                            ent["mfma_frac"] = round(bwd_flop_per_launch / (us * 1e-6) / 1e12 / PEAK_FP32_TFLOPS, 4)
                            ent["bound"] = "mfma"
                        else:
                            ent["bound"] = "latency (integer work)" if kname.startswith(("mb_", "mbw_", "plan_", "rowidx_", "partition_", "compute_rowidx")) else "hbm"
                        rows_.append(ent)
                    if rows_:
                        line["kernel_rooflines"] = rows_
            except Exception:  # noqa: BLE001
                pass
        if a2a is not None:
            line["all_to_all"] = a2a
        line["debug_knobs"] = debug_state  # (ttx_debug_state(): 0 = every test / ablation knob of the library at its default)
        if debug_state:
            line["INVALID"] = f"timed with libttx debug knobs set (mask {debug_state}, TTX_ALLOW_DEBUG=1): an ablation, not a measurement"
        if sharded and reported_ranks:
            # the ranks' own clocks around the LAST timed region of the reported mode (value uses the MAX, as the contract says)
            line["per_rank_ms_per_step"] = [round(t / args.steps * 1e3, 4) for t in reported_ranks]
        if note:
            line["note"] = note
        return line

    # ---- region 2 (the reported value): the same fwd+bwd step captured into a hipGraph (HIP streams and graphs
    # instead of per-launch host work) and replayed; every replay runs the full plan / forward / pool / backward /
    # apply kernel sequence on inputs resident in HBM.  One graph holds a whole round of the request batches (a
    # graph launch costs the host tens of microseconds, as much as a step's kernels take); single-step graphs serve
    # the remainder, so exactly K steps run either way.  N > 1: the exchanges go through RCCL directly
    # (ttx_sharded.DirectExchange: current stream, capturable; torch.distributed's own collectives are not).
    # Falls back to the eager timing if capture is unavailable.
    mode, regions, plain_regions = "eager", None, None
    degraded = None  # set when the reported mode is a fallback: the line says so and a multi-GPU run exits non-zero
    can_graph = not args.no_graph and (not wl["populate"] or ops._native_node() is not None)
    if sharded:
        can_graph = can_graph and ops._native_node() is not None and not os.environ.get("TTX_NO_DIRECT_RCCL")
    dog = None
    if can_graph:
        try:
            import ttx_graph

            if sharded:
                import threading

                def bail():  # a hung collective must not take the driver's slot with it -- and must not look like success
                    if rank == 0:
                        line = build_line("eager", eager_regions, {}, None,
                                          note="direct-RCCL graph region did not finish; eager torch.distributed result")
                        line["degraded"] = "watchdog: the captured direct-RCCL region hung; exit code 3"
                        print(json.dumps(line), flush=True)
                    print(f"[bench] rank {rank}: direct-RCCL graph region did not finish in time -- exiting with code 3",
                          file=sys.stderr, flush=True)
                    os._exit(3)

                dog = threading.Timer(float(os.environ.get("TTX_DIRECT_TIMEOUT", "240")), bail)
                dog.daemon = True
                mod.enable_direct_exchange()
                eager_steps(5)
                sync()
            # The captured round plans its batches ahead where the module supports it (one GPU; cache not live, or live over
            # one table: then the frequency updates, cache lookups, hit / miss partitions and miss plans): the lookup
            # prologues of the round's batches -- frequency update, bag rows, lookup plan: index work that depends on a
            # batch's indices only, not on the cores -- are enqueued up front in ONE launch (module.prefetch_many ->
            # ttx_lookup_prologue_multi; a prologue occupies 30 of the 256 CUs for ~12 us of dependent loads, ten of them
            # take about as long as one), the steps' forward / backward follow without them.  Same kernels' work, same
            # results (bit-identical, tested); every replay plans its ten batches again, inside the timed region.  The
            # plain round (every step's prologue in line) is timed beside it (`no_prefetch`).  `--prefetch next`: the
            # prologue of batch k+1 on a side stream under the backward of batch k instead (a forked branch in the graph:
            # measured slower than in line at this step size, the cross-stream edges cost more than the overlap saves).
            # N > 1 (sharded module, `--prefetch round`): the round's index exchange is planned ahead as well -- ONE all-to-all
            # carries the ten batches' lookups, the owners' prologues follow in one launch, every step is left with the
            # pooled exchange forward and the gradient exchange backward (ShardedTableBatchedTTEmbeddingBag.prefetch_many).
            # Planning ahead pays where the prologue is latency-bound (small batches); at 327k+ lookups per step it costs:
            # ten batches' plans and exchanged indices are written up front and read back from HBM instead of from the
            # caches they were just written through (cfg5 on one rank: 4.31 -> 4.89 ms/step), so large steps stay in line.
            pf_kw = {"fixed_pooling": POOL} if sharded else {}
            small = reqs[0][0].numel() <= 65536
            if not small:
                pipelined = False
            elif sharded:
                pipelined = args.prefetch == "round" and bool(mod.prefetch_many(reqs[:1], **pf_kw))
                if pipelined:
                    mod._planned.clear()
                    if mod.local is not None and getattr(mod.local, "_prefetched", None):
                        mod.local._prefetched.clear()
            else:
                pipelined = args.prefetch != "none" and hasattr(mod, "prefetch_many") and bool(
                    mod.prefetch_many(reqs[:1]) if args.prefetch == "round" else mod.prefetch(*reqs[0]))
                if pipelined:
                    mod._prefetched.clear()
            if pipelined:
                mk = ttx_graph.planned_round if args.prefetch == "round" else ttx_graph.pipelined_round
                round_fn = mk(mod, reqs, lambda out, k: out.backward(grad), **pf_kw)
            E.profile_reset()
            E.profile_mask(1 << E.PROF_BWD)  # the event pairs around the backward kernel become graph nodes
            g_round = ttx_graph.GraphedRound(round_fn, [()], warmup=3) if pipelined else ttx_graph.GraphedRound(step, reqs, warmup=3)
            E.profile_mask(0)
            singles = [ttx_graph.GraphedRound(step, [b], warmup=0) for b in reqs[:args.steps % iters]]

            def graph_steps(n):
                for _ in range(n // iters):
                    g_round.replay()
                for k in range(n % iters):
                    singles[k].replay()

            if dog is not None:
                dog.start()
            graph_steps(max(args.warmup, iters))
            regions = [timed(graph_steps, args.steps) for _ in range(max(1, args.repeats))]
            reported_ranks[:] = rank_regions[-1] if rank_regions else []
            mode = (("hipgraph+direct-rccl" + ("+prefetch-round" if pipelined else "")) if sharded
                    else (f"hipgraph+prefetch-{args.prefetch}" if pipelined else "hipgraph"))
            if pipelined:  # the same round without the overlap, for the record
                g_plain = ttx_graph.GraphedRound(step, reqs, warmup=1)
                pipelined_round_replay, g_round = g_round, g_plain
                graph_steps(iters)
                plain_regions = [timed(graph_steps, args.steps) for _ in range(max(1, min(3, args.repeats)))]
                g_round = pipelined_round_replay
            # the captured event pairs hold the times of the LAST replay of each of the round's steps: the live
            # duration of the dominant kernel inside the timed region, without the host's launch latency that an
            # eager event bracket picks up when the host, not the GPU, is the bottleneck
            try:
                n_g, ms_g = E.profile_read(E.PROF_BWD)
                if n_g > 0 and ms_g > 0:
                    n_bwd, ms_bwd, bwd_src = n_g, ms_g, "HIP events captured in the replayed graph"
            except RuntimeError:
                pass
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] graph capture unavailable ({type(ex).__name__}: {ex}); reporting the eager path", file=sys.stderr)
            degraded = f"graph capture failed ({type(ex).__name__}: {ex}); eager path reported"
            if sharded:
                mod.direct = None
                if hasattr(mod, "drop_planned"):
                    mod.drop_planned()
            regions = None
            torch.cuda.synchronize()
        finally:
            if dog is not None:
                dog.cancel()
    if regions is None:  # eager is the reported mode: repeat it like the graph region
        eager_regions += [timed(eager_steps, args.steps) for _ in range(max(0, args.repeats - 1))]
        regions = eager_regions
        reported_ranks[:] = rank_regions[-1] if rank_regions else []

    # N > 1: the two exchanges of a step in isolation (xGMI all-to-all bandwidth vs the link roofline), with the
    # step's own per-peer message sizes (uneven when the tables do not divide by the ranks)
    a2a = None
    if sharded:
        try:
            a2a = {}
            n_me = owned[rank]
            msgs = (("indices_in", [k * B_local * POOL for k in owned], [n_me * B_local * POOL] * world, torch.int64),
                    ("pooled_out", [n_me * B_local * D] * world, [k * B_local * D for k in owned], torch.float32))
            routes = [("", lambda o, i, osp, isp: dist.all_to_all_single(o, i, osp, isp))]
            if getattr(mod, "direct", None) is not None:
                routes.append(("_direct", lambda o, i, osp, isp: mod.direct.all_to_all(o, i, osp, isp)))
            for suffix, call in routes:
                for name, in_splits, out_splits, dtype in msgs:
                    src = torch.zeros(sum(in_splits), dtype=dtype, device=dev)
                    dst = torch.empty(sum(out_splits), dtype=dtype, device=dev)
                    for _ in range(5):
                        call(dst, src, out_splits, in_splits)
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        call(dst, src, out_splits, in_splits)
                    sync()
                    dt = (time.perf_counter() - t0) / 50
                    sent = src.element_size() * (sum(in_splits) - in_splits[rank])  # bytes this rank puts on the links
                    a2a[name + suffix] = {"bytes_on_links_this_rank": sent, "us": round(dt * 1e6, 1),
                                          "GB/s_per_rank": round(sent / dt / 1e9, 2),
                                          "frac_of_xgmi": round(sent / dt / 1e9 / (XGMI_GBS * max(1, min(world - 1, 7))), 4)}
            a2a["note"] = ("barrier-synchronised loop of 50 per message; '' = torch.distributed all_to_all_single over RCCL, "
                           "'_direct' = ncclAllToAll / ncclSend+ncclRecv group on the current stream; link roofline = 153 GB/s x "
                           "peers (MI355X_MICROARCH.md); messages this small are latency-bound")
        except Exception as ex:  # noqa: BLE001
            a2a = {"error": f"{type(ex).__name__}: {ex}"}

    # second, untimed pass: per-kernel breakdown (all kernel slots bracketed)
    E.profile_reset()
    E.profile_enable(0x3F)
    eager_steps(min(args.steps, 50))
    sync()
    E.profile_enable(0)
    names = ["fwd_contract", "bwd_contract", "reduce_apply", "plan", "bag_pool", "cache_gather"]
    breakdown, launches = {}, {}
    for w, nm in enumerate(names):
        n, ms = E.profile_read(w)
        if n:
            breakdown[nm + "_us"] = round(ms / n * 1e3, 2)
            launches[nm] = n

    if rank == 0:
        line = build_line(mode, regions, breakdown, a2a)
        if wl["populate"] and "cache_gather_us" not in breakdown and "bag_pool_us" in breakdown and hit_rate is not None:
            # (round 4) the gather runs in the pooling launch (ttx_tt_forward_cached): the roofline of that launch -- the cache
            # rows as below plus the contraction's rows of the misses (4*D + 8 bytes each)
            breakdown = dict(breakdown)
            breakdown["cache_gather_us"] = breakdown["bag_pool_us"]
            fused_gather = True
        else:
            fused_gather = False
        if wl["populate"] and "cache_gather_us" in breakdown and hit_rate is not None:
            # cache-hit gather (a11): 4*D + 4 + 8 bytes per cached lookup + 4*D per bag, HBM-bound
            cached = hit_rate * nnz_step_total
            bytes_ = cached * (4 * D + 12) + B_GLOBAL * 4 * D
            if fused_gather:
                bytes_ += (nnz_step_total - cached) * (4 * D + 8)
            gbs = bytes_ / (breakdown["cache_gather_us"] * 1e-6) / 1e9
            line["cache_gather_roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                                             "frac": round(gbs / (PEAK_HBM_TBS * 1e3), 4), "cached_lookups_per_launch": round(cached),
                                             "bytes_per_launch": round(bytes_), "avg_us": breakdown["cache_gather_us"],
                                             "timed_by": "HIP events around each launch (includes launch gap); rocprof figures: profiles/",
                                             "kernel": ("pool4_small_cached_kernel (bag sums of the contraction's rows and of the cache's rows, "
                                                        "one launch)" if fused_gather else "cache_forward4_kernel")}
            # the same kernel as rocprofv3 --kernel-trace --stats timed it on THIS build (scripts/kprof.sh -> profiles/rocprof_kernels.json)
            rk = os.path.join(ROOT, "profiles", "rocprof_kernels.json")
            try:
                j = json.load(open(rk))
                ks = j["workloads"].get(args.workload, {}) if j.get("source_hash") == source_hash() else {}
                us = next((v["avg_us"] for k, v in ks.items() if k.startswith("pool4_small_cached" if fused_gather else "cache_forward")), None)
            except Exception:  # noqa: BLE001
                us = None
            if us:
                g2 = bytes_ / (us * 1e-6) / 1e9
                line["cache_gather_roofline"].update({"rocprof_avg_us": us, "achieved_rocprof": round(g2, 1),
                                                      "frac_rocprof": round(g2 / (PEAK_HBM_TBS * 1e3), 4)})
            else:
                line["cache_gather_roofline"]["rocprof_avg_us"] = None
        if hit_rate is not None:
            line["cache_hit_rate"] = round(hit_rate, 4)
        if not sharded and not args.no_cpu_baseline and ntab == 1:
            line["cpu_baseline"] = cpu_baseline(reqs_np, cores_np, d_out_np, Q_SHAPES, RANKS, B_GLOBAL)
        if args.run_baseline and not sharded and ntab == 1:
            line["dense_embedding_bag"] = dense_baseline(E_, D, reqs, grad, args.steps, args.warmup)
        if degraded:
            line["degraded"] = degraded
        if args.workload == "cfg2" and not sharded and not args.no_secondary:
            # (round 6) every BASELINE.json config under the driver's clock: besides the default line's configs[1], one rank's
            # share of configs[4] (the regime where the contraction kernels are the bound), configs[3] (fused Adagrad at ranks 64,
            # D = 128: its own roofline) and configs[2] (the row cache live on a Zipf stream: hit rate, cache_gather_roofline).
            # A process each: a second module beside the first would perturb both.
            line["secondary"] = [secondary_record("cfg5shard", 40, 3), secondary_record("cfg4", 100, 3),
                                 secondary_record("cfg3", 100, 3), secondary_record("cfg5", 10, 3)]
            # (configs[4] whole on ONE GPU -- 26 tables, 2.13 M lookups per step -- is the base of the predicted 8-GPU speed-up)
            try:
                pred = line["secondary"][0].get("predicted_cfg5_on_8_gpus")
                if pred and "ms_per_step" in line["secondary"][3]:
                    pred["one_gpu_ms_per_step"] = line["secondary"][3]["ms_per_step"]
                    pred["predicted_speedup_over_one_gpu"] = round(line["secondary"][3]["ms_per_step"] / pred["ms_per_step"], 2)
            except Exception:  # noqa: BLE001
                pass
            if "dense_embedding_bag" not in line and ntab == 1:
                # the reference benchmark's --run-baseline leg (tt_embeddings_benchmark.py:195-211): the dense table the cores replace
                try:
                    line["dense_embedding_bag"] = dense_baseline(E_, D, reqs, grad, args.steps, args.warmup)
                    line["dense_embedding_bag"]["tt_speedup"] = round(line["dense_embedding_bag"]["ms_per_step"] / line["eager_ms_per_step"], 3)
                    line["dense_embedding_bag"]["tt_speedup_what"] = "dense eager ms/step over the TT module's eager ms/step (the same loop form)"
                except Exception as ex:  # noqa: BLE001 -- a 2.8 GB table that does not fit must not take the line with it
                    line["dense_embedding_bag"] = {"error": f"{type(ex).__name__}: {ex}"}
        if not (sharded and world > 1):
            print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        sys.stdout.flush()
        # Tear the communicators down like a well-behaved job -- under a timeout: ncclCommDestroy / destroy_process_group
        # were seen to hang on this stack (one rank, sandboxed box).  A teardown that hangs is reported on stderr; the
        # measurement above is complete either way.  A degraded multi-GPU run (fallback mode) exits non-zero.
        import threading

        done = threading.Event()

        def teardown():
            try:
                direct = getattr(mod, "direct", None)
                if direct is not None and hasattr(direct, "close"):
                    direct.close()
                dist.destroy_process_group()
            except Exception as ex:  # noqa: BLE001
                print(f"[bench] rank {rank}: teardown raised {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)
            done.set()

        th = threading.Thread(target=teardown, daemon=True)
        th.start()
        if not done.wait(float(os.environ.get("TTX_TEARDOWN_TIMEOUT", "10"))):
            print(f"[bench] rank {rank}: communicator teardown did not finish in time; leaving it to process exit",
                  file=sys.stderr, flush=True)
        rc = 4 if (degraded and world > 1) else 0
        if rank == 0 and world > 1:
            # N > 1: the ONE line is printed here, after the communicators are gone and the other ranks have left rank 0's GPU alone,
            # with the one-GPU value of the same workload beside it (the driver computes the scaling efficiency itself; this is the
            # same-box, same-job denominator for anyone reading the line alone)
            if not args.no_n1:
                line["n1"] = n1_record(args.workload, args.steps, args.warmup, local_rank)
                if "value" in line["n1"] and line["n1"]["value"] > 0:
                    # weak scaling: per-GPU work fixed -> ideal = N x n1; strong: total work fixed -> ideal = N x n1 as well (value is the
                    # whole-job rate either way)
                    line["n1"]["value_over_n_times_n1"] = round(line["value"] / (world * line["n1"]["value"]), 4)
            print(json.dumps(line), flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)


if __name__ == "__main__":
    main()
