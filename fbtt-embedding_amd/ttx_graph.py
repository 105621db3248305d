"""hipGraph capture of training steps (not in the reference).

At the benchmark batch a fwd+bwd step is five kernels of ~10 us each; launching them one by one costs the host
more than they take to run, and even one hipGraphLaunch per step costs about as much as the step's kernels.
`GraphedRound` captures `step(*batch)` for a list of batches whose tensors stay resident (static input
buffers: copy new data into them between replays) into ONE graph; `replay()` runs the whole round.
Works with the C++ lookup node (tt_embeddings_ops falls back to the ctypes route, which also captures as long
as the cache is not live).  bench.py times exactly this."""
from typing import Callable, Sequence

import torch


def planned_round(module, batches: Sequence[tuple], backward: Callable, **forward_kw) -> Callable:
    """A round of training steps over `batches` whose lookup prologues (frequency update, bag rows, lookup plan: index
    work that depends on a batch's indices only) are all enqueued up front in one launch (`module.prefetch_many`), the
    steps' forward / backward following without them.  `forward_kw` goes to both calls (the sharded module's
    `fixed_pooling=L`: its index exchange is planned ahead as well).  Returns a zero-argument callable for GraphedRound."""
    batches = list(batches)

    def run() -> None:
        module.prefetch_many(batches, **forward_kw)
        for k, (i, o) in enumerate(batches):
            backward(module(i, o, **forward_kw), k)

    return run


def pipelined_round(module, batches: Sequence[tuple], backward: Callable) -> Callable:
    """A round of training steps over `batches` in which the lookup prologue of batch k+1 (`module.prefetch`: frequency
    update, bag rows, lookup plan -- index work that does not depend on the cores) is enqueued on a side stream before
    the backward of batch k, so that the two overlap.  `backward(out, k)` runs the step's backward (e.g.
    `lambda out, k: out.backward(grad)`).  Returns a zero-argument callable for GraphedRound (or to call eagerly)."""
    batches = list(batches)

    def run() -> None:
        for k, (i, o) in enumerate(batches):
            out = module(i, o)
            if k + 1 < len(batches):
                module.prefetch(*batches[k + 1])
            backward(out, k)

    return run


class GraphedRound:
    def __init__(self, step: Callable, batches: Sequence[tuple], warmup: int = 3) -> None:
        """step(*batch) for every batch -- or, with batches == [()], one zero-argument round function such as
        pipelined_round(...)."""
        assert len(batches) > 0
        self.batches = list(batches)  # keep the captured tensors alive
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):  # allocator / lazy-init warm-up outside the capture
            for k in range(warmup):
                step(*self.batches[k % len(self.batches)])
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()
        try:
            import tt_embeddings as _E

            _E._ws_cache.clear()  # ctypes route: the graph must own its workspaces
        except ImportError:
            _E = None
        self.graph = torch.cuda.CUDAGraph()
        # "thread_local": only this thread's calls are held to the capture rules.  With a process group alive,
        # torch.distributed's watchdog thread polls its events all the time; under the default ("global") mode
        # such a poll during the capture fails and the watchdog aborts the process.
        with torch.cuda.graph(self.graph, stream=self._stream, capture_error_mode="thread_local"):
            for b in self.batches:
                step(*b)
        if _E is not None:
            _E._ws_cache.clear()

    def replay(self) -> None:
        self.graph.replay()

    def __len__(self) -> int:
        return len(self.batches)


class GraphedStep:
    """A training step behind static input buffers: the drop-in way to the replayed number (round 4).

        step = ttx_graph.GraphedStep(lambda idx, off, grad: emb(idx, off).backward(grad), (idx0, off0, grad0))
        for idx, off, grad in loader:      # tensors of the SAME shapes and dtypes as the example
            step(idx, off, grad)           # copies them into the static buffers, replays the graph

    `fn(*tensors)` is captured once over copies of `example` (after `warmup` eager calls on a side stream); every later call
    costs one `copy_` per input and one hipGraphLaunch instead of the step's Python + launches (0.086 vs 0.045 ms at the
    benchmark config).  The shapes are part of the graph: a batch with another number of lookups needs its own GraphedStep (or
    padding to a common length with lookups of an all-zero bag weight).  The fused optimizers update the TT cores inside the
    graph; anything `fn` returns is ignored (read results from tensors `fn` writes into, e.g. a preallocated loss buffer)."""

    def __init__(self, fn: Callable, example: Sequence[torch.Tensor], warmup: int = 3) -> None:
        self.static = [t.clone() for t in example]
        self._round = GraphedRound(lambda *a: fn(*a), [tuple(self.static)], warmup=warmup)

    def __call__(self, *tensors: torch.Tensor) -> None:
        assert len(tensors) == len(self.static), f"{len(self.static)} inputs were captured"
        for dst, src in zip(self.static, tensors):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"GraphedStep: captured {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._round.replay()
