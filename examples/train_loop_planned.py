#!/usr/bin/env python3
"""A training loop over one TT-compressed embedding table that uses the whole life cycle of the module:

    python examples/train_loop_planned.py          # cuda:0

  1. warm-up steps: the LFU table counts the indices it sees (the reference's `warmup` phase),
  2. `cache_populate()`: the most frequent rows are decompressed into the row cache, lookups of cached rows
     become gathers from then on,
  3. steady state, a round of queued batches at a time: `prefetch_many(batches)` plans the round's lookups ahead
     (frequency update, cache lookup, hit / miss partition and the miss plans of all batches in three launches:
     index work that depends on the batches only, not on the weights), then the steps follow without it,
  4. the same round captured in a hipGraph and replayed (ttx_graph.GraphedRound): one graph launch per round.

Nothing here differs in RESULT from calling `emb(indices, offsets)` step by step; only where the index work runs."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))


def zipf_batches(n, B, L, E, seed):
    rs = np.random.RandomState(seed)
    off = torch.arange(0, B * L + 1, L, dtype=torch.int64)
    return [(torch.from_numpy((rs.zipf(1.2, size=B * L).astype(np.int64)) % E), off.clone()) for _ in range(n)]


def main():
    import tt_embeddings_ops as ops
    import ttx_graph

    dev = torch.device("cuda:0")
    E, D, B, L = 11_000_000, 64, 512, 20
    emb = ops.TTEmbeddingBag(E, D, [32, 32], [200, 220, 250], [4, 4, 4], sparse=True, optimizer=ops.OptimType.SGD,
                             learning_rate=0.05, use_cache=True, cache_size=1 << 16, hashtbl_size=1 << 20,
                             weight_dist="uniform", device=dev)
    head = torch.nn.Linear(D, 1).to(dev)
    opt = torch.optim.SGD(head.parameters(), lr=0.05)  # (the TT cores are updated by the lookup's fused backward)
    target = torch.zeros(B, device=dev)

    def step(indices, offsets):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(head(emb(indices, offsets)).squeeze(1), target)
        loss.backward()
        opt.step()
        return loss

    # 1. warm-up: plain steps; the frequency table fills
    for i, o in zipf_batches(20, B, L, E, seed=1):
        step(i.to(dev), o.to(dev))
    # 2. the cache goes live
    emb.cache_populate()
    # 3. steady state, a round of 8 queued batches at a time
    queue = [(i.to(dev), o.to(dev)) for i, o in zipf_batches(8, B, L, E, seed=2)]
    for _ in range(3):
        planned = emb.prefetch_many(queue)  # False where the module cannot plan ahead (then every step plans itself)
        for i, o in queue:
            loss = step(i, o)
    torch.cuda.synchronize()
    print(f"steady state, prologues planned ahead: {planned}; loss {float(loss.detach()):.6f}")

    # 4. the same round as ONE hipGraph (static batch tensors: copy new data into them between replays).
    #    Only the embedding's forward + backward is captured here -- a torch optimizer captures too, but needs its own
    #    capturable settings; the lookup's fused optimizer is part of its backward.
    grad = torch.full((B, D), 1e-3, device=dev)
    rnd = ttx_graph.GraphedRound(ttx_graph.planned_round(emb, queue, lambda out, k: out.backward(grad)), [()])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        rnd.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (50 * len(queue))
    print(f"captured round of {len(queue)} cache-live steps: {dt * 1e3:.4f} ms per step (fwd + bwd + fused SGD)")


if __name__ == "__main__":
    main()
