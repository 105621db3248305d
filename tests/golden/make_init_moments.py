#!/usr/bin/env python3
"""tests/golden/init_moments.json: per-core statistics of the REFERENCE's five `reset_parameters` initialisers
(tt_embeddings_ops.py:613-792) under fixed seeds.

Build container only (needs /root/reference): imports the reference's tt_embeddings_ops.py with an empty stand-in
for its CUDA extension module and torch.cuda patched so the classes instantiate on the CPU;
only numbers are written.  tests/test_module_cpu.py::test_initialisers_match_reference_moments seeds the same three
generators (torch / numpy / random) and compares this repository's own initialisers with them."""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("tt_embeddings", types.ModuleType("tt_embeddings"))
sys.path.insert(0, "/root/reference")
torch.cuda.is_available = lambda: True          # (the reference asserts a GPU, ops.py:454)
torch.cuda.current_device = lambda: "cpu"
import tt_embeddings_ops as ref  # noqa: E402

CASES = [
    dict(E=11000, D=64, ranks=[16, 16], p=[20, 22, 25], q=[4, 4, 4]),
    dict(E=1000000, D=64, ranks=[12, 14], p=[100, 100, 100], q=[4, 4, 4]),
]
DISTS = ["uniform", "naive-uniform", "normal", "approx-normal", "approx-uniform"]
SEED = 2024


def stats(x):
    x = x.detach().double().numpy().ravel()
    return dict(mean=float(x.mean()), std=float(x.std()), min=float(x.min()), max=float(x.max()),
                abs_mean=float(np.abs(x).mean()), first=[float(v) for v in x[:4]])


def main():
    out = {"seed": SEED, "cases": []}
    for c in CASES:
        for dist in DISTS:
            torch.manual_seed(SEED)
            np.random.seed(SEED)
            random.seed(SEED)
            m = ref.TTEmbeddingBag(c["E"], c["D"], c["ranks"], c["p"], c["q"], sparse=False, use_cache=False, weight_dist=dist)
            out["cases"].append(dict(cfg=c, dist=dist, cores=[stats(t) for t in m.tt_cores],
                                     full=stats(m.full_weight()) if c["E"] <= 20000 else None))
            print(c["E"], dist, [round(s["std"], 6) for s in out["cases"][-1]["cores"]])
    json.dump(out, open(os.path.join(HERE, "init_moments.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
