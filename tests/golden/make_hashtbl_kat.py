"""Regenerates tests/golden/hashtbl_kat.json from the REFERENCE's own code: hashtbl_cuda_utils.cuh's hash,
hashtbl_insert<int64,int64,true> and hashtbl_find, compiled for gfx950 by `make -C oracle refdev`
(oracle/_ref/libcacheref.so) and run on the GPU.  Needs a GPU: run on the GPU box (`make -C oracle kat`);
tests/test_refdev_gpu.py::test_kat_json_is_what_the_reference_code_produces checks the committed file against
the same build on every GPU run.  The hash values are also checked on the CPU against the host build of the
same reference lines (oracle/_ref/libhashref.so) by tests/test_oracle_golden.py."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refdev_lib as R  # noqa: E402

DEV = "cuda:0"
SIZES = [16, 1000, 1048576, 11000000]
HASH_KEYS = [0, 1, 2, 9, 12345, 10999999, 4294967301, -1]
INSERT_KEYS = [1, 2, 4, 5, 4, 3, 2, 9, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
FIND_KEYS = [4, 9, 100, -1]


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def main():
    old = json.load(open(os.path.join(HERE, "hashtbl_kat.json")))
    keys = np.array(HASH_KEYS, dtype=np.int64)
    per_size = [R.hash64(t(keys), H) for H in SIZES]
    kat = {
        "_provenance": "outputs of the reference's own hashtbl_cuda_utils.cuh (hash :44-98, insert/find :100-154) compiled for "
                       "gfx950 by oracle/Makefile target refdev and run on MI355X by tests/golden/make_hashtbl_kat.py; first "
                       "captured in SURVEY.md Appendix C from a host build of the same lines",
        "sizes": SIZES,
        "hash64": {str(k): [int(per_size[j][i]) for j in range(len(SIZES))] for i, k in enumerate(HASH_KEYS)},
        "hash64_raw": old["hash64_raw"],  # (pre-modulo values: host build only, test_oracle_golden.py)
        "hash32_12345_1000": old["hash32_12345_1000"],
    }
    H = 16
    dk, df = t(np.full(H, -1, dtype=np.int64)), t(np.zeros(H, dtype=np.int64))
    ret = R.insert_seq(t(np.array(INSERT_KEYS, dtype=np.int64)), dk, df).cpu().numpy()
    found = R.find(t(np.array(FIND_KEYS, dtype=np.int64)), dk).cpu().numpy()
    kat["insert"] = {"size": H, "keys": INSERT_KEYS, "returns": ret.tolist(), "final_keys": dk.cpu().numpy().tolist(),
                     "final_freqs": df.cpu().numpy().tolist(), "find": {str(k): int(v) for k, v in zip(FIND_KEYS, found)}}
    out = os.path.join(HERE, "hashtbl_kat.json")
    same = {k: kat[k] for k in kat if k != "_provenance"} == {k: old[k] for k in old if k != "_provenance"}
    json.dump(kat, open(out, "w"), indent=2)
    print(f"wrote {out}; identical to the committed vectors: {same}")


if __name__ == "__main__":
    main()
