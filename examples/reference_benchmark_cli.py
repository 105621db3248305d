"""The reference benchmark's command line on this library (tt_embeddings_benchmark.py:123-211 names the flags and the three
printed quantities; the code below is this repository's own).

    python examples/reference_benchmark_cli.py                       # cfg2 of BASELINE.json: the reference's defaults
    python examples/reference_benchmark_cli.py --optimizer adagrad --q-shapes 4,4,8 --ranks 64,64
    python examples/reference_benchmark_cli.py --run-baseline        # nn.EmbeddingBag(sparse=True) on the same requests

What is timed is the reference's loop, unchanged: `iters` requests generated up front, one untimed pass over them, then one pass
of `tt_emb(indices, offsets).backward(grad_output)` per request between two device events, divided by the number of requests
(tt_embeddings_benchmark.py:94-108) -- eager, no graph, no planning ahead.  Printed like the reference prints it: time per
lookup, "GFLOPS" by the reference's formula (which multiplies the FLOP of ONE request by `iters` and divides by the time of one:
ten times the true rate at the default `--iters 10`) with the true figure beside it, and the row bandwidth.
bench.py is the measured harness of this repository (graph replay, roofline, CPU baseline); this script is the like-for-like."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))
from tt_embeddings_ops import OptimType, TTEmbeddingBag  # noqa: E402


def ints(text):
    vals = [int(v) for v in text.split(",")]
    if not vals or min(vals) <= 0:
        raise argparse.ArgumentTypeError(f"positive integers separated by commas, got {text!r}")
    return vals


def make_requests(iters, B, L, E, dtype, device, seed=0):
    """`iters` x (indices [B L] uniform over the table, offsets [B + 1] of fixed-length bags), resident on the device"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    offsets = torch.arange(0, B * L + 1, L, dtype=dtype, device=device)
    return [(torch.randint(0, E, (B * L,), generator=g, dtype=dtype).to(device), offsets) for _ in range(iters)]


def seconds_per_request(requests, step):
    for indices, offsets in requests:  # untimed pass (allocator, first-touch, the frequency table's first inserts)
        step(indices, offsets)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for indices, offsets in requests:
        step(indices, offsets)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e-3 / len(requests)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pooling-factor", type=int, default=20)
    ap.add_argument("--p-shapes", type=ints, default=[200, 220, 250])
    ap.add_argument("--q-shapes", type=ints, default=[4, 4, 4])
    ap.add_argument("--ranks", type=ints, default=[32, 32])
    ap.add_argument("--int32-index", action="store_true", help="32-bit indices (the reference's --long-index defaults to on)")
    ap.add_argument("--dense", action="store_true", help="dense core gradients instead of the fused optimizer (the reference's --sparse defaults to on)")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adagrad"])
    ap.add_argument("--run-baseline", action="store_true")
    ap.add_argument("--direct-backward", action="store_true",
                    help="opt in to tt_embeddings_ops.enable_direct_backward(): backward() of the lookup's own output skips autograd's engine")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    E, D = int(np.prod(a.p_shapes)), int(np.prod(a.q_shapes))
    B, L = a.batch_size, a.pooling_factor
    nnz = B * L
    reqs = make_requests(a.iters, B, L, E, torch.int32 if a.int32_index else torch.int64, dev)
    q, r = a.q_shapes, [1] + a.ranks + [1]
    # FLOP of one lookup's forward, core after core, left to right (any number of cores; the reference's formula is the T = 3 case)
    per_lookup, rows = 0, 1
    for t in range(len(q)):
        rows *= q[t]
        if t > 0:
            per_lookup += 2 * (rows // q[t]) * r[t] * q[t] * r[t + 1]
    flop_fwd = float(nnz) * per_lookup
    tt = TTEmbeddingBag(num_embeddings=E, embedding_dim=D, tt_p_shapes=a.p_shapes, tt_q_shapes=q, tt_ranks=a.ranks, sparse=not a.dense,
                        optimizer=OptimType.SGD if a.optimizer == "sgd" else OptimType.EXACT_ADAGRAD, use_cache=True).to(dev)
    grad = torch.rand(B, D, device=dev) * 0.1
    if a.direct_backward:
        import tt_embeddings_ops

        tt_embeddings_ops.enable_direct_backward()
    t = seconds_per_request(reqs, lambda i, o: tt(i, o).backward(grad))
    print(f"B: {B}, E: {E}, D: {D}, nnz: {nnz}, p: {a.p_shapes}, q: {q}, ranks: {a.ranks}, optimizer: {a.optimizer}, sparse: {not a.dense}")
    print(f"TTEmbeddingBag FWD-BWD time/nnz: {t / nnz * 1e6:.4f} usecs ({t * 1e6:.1f} us per request), "
          f"GFLOPS by the reference's formula (x iters = {a.iters}): {3.0 * flop_fwd * a.iters / t / 1e9:.1f}, "
          f"true GFLOPS: {3.0 * flop_fwd / t / 1e9:.1f}, BW (rows, x iters): {3.0 * 4.0 * nnz * D * a.iters / t / 1e9:.1f} GB/s")
    if a.run_baseline:
        emb = torch.nn.EmbeddingBag(E, D, sparse=True, mode="sum", include_last_offset=True).to(dev)
        opt = torch.optim.SGD(emb.parameters(), lr=0.1)

        def step(i, o):
            opt.zero_grad(set_to_none=True)
            emb(i.long(), o.long()).backward(grad)
            opt.step()  # (the TT module's backward includes its optimizer step: so does this)

        t = seconds_per_request(reqs, step)
        print(f"EmbeddingBag FWD-BWD(+SGD) time/nnz: {t / nnz * 1e6:.4f} usecs ({t * 1e6:.1f} us per request), table {E * D * 4 / 2**30:.2f} GiB, "
              f"BW (rows, x iters): {3.0 * 4.0 * nnz * D * a.iters / t / 1e9:.1f} GB/s")


if __name__ == "__main__":
    main()
