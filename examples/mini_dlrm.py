#!/usr/bin/env python3
"""A DLRM-shaped toy model (bottom MLP over dense features, one embedding bag per sparse feature, dot
interaction, top MLP) whose sparse side is either nn.EmbeddingBag or the TT-compressed drop-in:

    python examples/mini_dlrm.py            # trains a few steps with TTEmbeddingBag on cuda:0

The embedding call form is DLRM's: `emb(indices, offsets)` with offsets holding only the bag starts
(nn.EmbeddingBag's default), which TTEmbeddingBag accepts with include_last_offset=False."""
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbtt-embedding_amd"))


class MiniDLRM(nn.Module):
    def __init__(self, embeddings, dense_in=13, d=64):
        super().__init__()
        self.emb = nn.ModuleList(embeddings)
        self.bot = nn.Sequential(nn.Linear(dense_in, 128), nn.ReLU(), nn.Linear(128, d), nn.ReLU())
        n = len(embeddings) + 1
        self.top = nn.Sequential(nn.Linear(d + n * (n - 1) // 2, 128), nn.ReLU(), nn.Linear(128, 1))

    def forward(self, dense, sparse):  # sparse: list of (indices, offsets) per feature
        x = self.bot(dense)
        feats = [x] + [e(i, o) for e, (i, o) in zip(self.emb, sparse)]
        T = torch.stack(feats, dim=1)                       # [B, n, d]
        Z = torch.bmm(T, T.transpose(1, 2))                 # pairwise dots
        iu = torch.triu_indices(T.size(1), T.size(1), offset=1, device=T.device)
        return self.top(torch.cat([x, Z[:, iu[0], iu[1]]], dim=1)).squeeze(1)


def tt_embeddings(num_features, device, sparse=True):
    import tt_embeddings_ops as ops

    return [ops.TTEmbeddingBag(11_000_000, 64, [32, 32], [200, 220, 250], [4, 4, 4], sparse=sparse,
                               optimizer=ops.OptimType.SGD, learning_rate=0.05, use_cache=False,
                               weight_dist="approx-normal", include_last_offset=False, device=device)
            for _ in range(num_features)]


def batch(B, num_features, device, seed):
    g = torch.Generator().manual_seed(seed)
    dense = torch.rand(B, 13, generator=g).to(device)
    sparse = []
    for _ in range(num_features):
        lengths = torch.randint(1, 6, (B,), generator=g)
        offsets = torch.cat([torch.zeros(1, dtype=torch.int64), lengths.cumsum(0)[:-1]])
        indices = torch.randint(0, 11_000_000, (int(lengths.sum()),), generator=g)
        sparse.append((indices.to(device), offsets.to(device)))
    label = (dense.sum(1) > 6.5).float()
    return dense, sparse, label


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = MiniDLRM(tt_embeddings(4, dev)).to(dev)
    dense_params = [p for n, p in model.named_parameters() if not n.startswith("emb.")]
    opt = torch.optim.SGD(dense_params, lr=0.05)  # the TT cores are updated by their fused optimizer in backward
    for step in range(20):
        dense, sparse, label = batch(256, 4, dev, step % 4)
        loss = nn.functional.binary_cross_entropy_with_logits(model(dense, sparse), label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 5 == 0:
            print(f"step {step}: loss {loss.item():.4f}")
