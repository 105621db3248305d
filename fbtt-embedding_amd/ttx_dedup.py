"""Duplicate indices share their contraction (opt-in wrapper around a one-table `TTEmbeddingBag`).

The lookup plan groups lookups by core slice, not by full index, so two lookups of the SAME index are
contracted twice.  Under a uniform stream that is nothing (≈5 duplicates among cfg2's 10,240 lookups); under
a skewed stream without a populated cache it is most of the work (Zipf 1.2: a sixth of a batch is one index).
`DedupTTEmbeddingBag` contracts every distinct index of a batch once:

    unique indices (torch.unique: a device sort + one host read-back of their number)
      -> the wrapped module with ONE bag per distinct index        -> rows [U, D]   (TT contraction, U <= nnz)
      -> F.embedding_bag(inverse, rows, offsets, mode="sum")       -> [B, D]        (gather + pool, torch)

and the backward mirrors it: torch scatters the bag gradients onto the distinct rows (index_add), the
wrapped module's fused backward then sees one gradient row per distinct index.  Mathematically identical
to the plain module (a row's gradient is the sum over its occurrences either way); the fp32 summation
order differs.  It costs a sort, a synchronisation and torch's gather/pool kernels per step, and MEASURED
it does not pay at the benchmark geometry (scripts/bench_dedup.py, eager fwd+bwd+SGD, no cache, MI355X):
    B=512    Zipf 1.2 (2,896 distinct of 10,240)   plain 0.093 ms/step   dedup 0.448
    B=16384  Zipf 1.2 (52k distinct of 327,680)    plain 1.23            dedup 2.25
    B=16384  uniform                               plain 1.46            dedup 2.11
-- the contraction is no longer what a skewed step spends its time on (hot slices are reduced by several
work-groups, equal hash keys and cache rows are combined), and the reference's own answer to skew, the row
cache, removes the repeats before they reach it (cfg3: 0.063 ms/step).  Kept as a tested reference point
for the "duplicate rows share contraction work" design question, not as a recommended path."""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


class DedupTTEmbeddingBag(nn.Module):
    def __init__(self, bag: nn.Module) -> None:
        super().__init__()
        assert getattr(bag, "num_tables", 1) == 1, "one table"
        self.bag = bag
        self.last_unique = 0  # distinct indices of the last batch (for reporting)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor,
                per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        include_last = bool(getattr(self.bag, "include_last_offset", True))
        uniq, inverse = torch.unique(indices.long(), sorted=True, return_inverse=True)
        U = int(uniq.numel())
        self.last_unique = U
        one_each = torch.arange(U + 1 if include_last else U, dtype=torch.int64, device=indices.device)
        rows = self.bag(uniq, one_each)  # [U, D]: the wrapped module, one bag per distinct index
        return F.embedding_bag(inverse, rows, offsets.long(), mode="sum", include_last_offset=include_last,
                               per_sample_weights=per_sample_weights)
