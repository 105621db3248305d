"""bench.py's command line as the driver uses it: `python bench.py --gpus N` must run by itself."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_more_gpus_than_the_box_has_is_refused_with_the_device_count():
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = have + 1 if have >= 1 else 2
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(want)], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 2, out.stderr[-2000:]
    assert f"needs {want} visible GPUs" in out.stderr and f"shows {have}" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # no bench line from a refused run


def test_world_size_that_contradicts_gpus_is_refused():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_sharded_code_path_runs_as_a_subprocess_and_prints_one_line():
    """the N > 1 path (process group over RCCL, sharded module, direct all-to-all, captured round) with one rank"""
    env = _env()
    env.setdefault("TTX_DIRECT_TIMEOUT", "200")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-sharded", "--steps", "4", "--warmup", "2", "--repeats", "1",
                          "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["value"] > 0 and j["unit"] == "GFLOP/s"
    assert "degraded" not in j, j.get("degraded")
    assert j["roofline"]["frac"] > 0 and j["roofline"]["peak"] == 157.3
    a2a = j["all_to_all"]
    assert "error" not in a2a and a2a["pooled_out"]["us"] > 0 and "frac_of_xgmi" in a2a["indices_in"]
    assert j["per_rank_ms_per_step"] and len(j["per_rank_ms_per_step"]) == 1


@pytest.mark.gpu
def test_check_indices_is_opt_in_and_names_the_offender():
    """TTX_CHECK_INDICES=1 (round 4 verdict: no index-range check anywhere, even opt-in): an index >= prod(p) -- which the plan
    kernels otherwise clamp silently and the reference reads out of bounds with (tt_embeddings_cuda.cu:795-799) -- makes the call
    fail with TTX_EINVAL naming the lookup; valid batches pass; without the variable nothing is checked (the clamp applies)."""
    prog = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "fbtt-embedding_amd"))
import tt_embeddings_ops as ops
dev = torch.device("cuda:0")
m = ops.TTEmbeddingBag(720, 64, [16, 16], [8, 9, 10], [4, 4, 4], sparse=True, optimizer=ops.OptimType.SGD, use_cache=False,
                       weight_dist="uniform", device=dev)
off = torch.arange(0, 41, 4, device=dev)
good = torch.randint(0, 720, (40,), device=dev)
m(good, off).sum().item()
bad = good.clone(); bad[17] = 720
try:
    m(bad, off).sum().item()
    print("NO-ERROR")
except RuntimeError as ex:
    print("ERROR:", ex)
neg = good.clone(); neg[3] = -5
try:
    m(neg, off).sum().item()
    print("NO-ERROR")
except RuntimeError as ex:
    print("ERROR:", ex)
m(good, off).sum().item()
print("DONE")
''' % ROOT
    env = _env()
    env["TTX_CHECK_INDICES"] = "1"
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith(("ERROR", "NO-ERROR", "DONE"))]
    assert len(lines) == 3 and lines[2] == "DONE", out.stdout[-2000:] + out.stderr[-3000:]
    assert "TTX_CHECK_INDICES" in lines[0] and "lookup 17" in lines[0] and "index 720" in lines[0], lines[0]
    assert "TTX_CHECK_INDICES" in lines[1] and "lookup 3" in lines[1] and "index -5" in lines[1], lines[1]
    env.pop("TTX_CHECK_INDICES")
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith(("ERROR", "NO-ERROR", "DONE"))]
    assert lines == ["NO-ERROR", "NO-ERROR", "DONE"], out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_the_reference_benchmarks_command_line_runs():
    """examples/reference_benchmark_cli.py: the reference benchmark's flags and loop form (tt_embeddings_benchmark.py:123-211) on
    this library -- a small table, both optimizers, the nn.EmbeddingBag baseline: exit 0 and the three printed quantities."""
    import re

    script = os.path.join(ROOT, "examples", "reference_benchmark_cli.py")
    for extra in (["--run-baseline"], ["--optimizer", "adagrad", "--int32-index"], ["--dense", "--q-shapes", "2,4,8", "--ranks", "16,8"]):
        out = subprocess.run([sys.executable, script, "--batch-size", "64", "--iters", "4", "--pooling-factor", "5", "--p-shapes", "20,22,25"]
                             + extra, capture_output=True, text=True, timeout=600, env=_env())
        assert out.returncode == 0, out.stderr[-2000:]
        m = re.search(r"TTEmbeddingBag FWD-BWD time/nnz: +([0-9.]+) usecs.*true GFLOPS: +([0-9.]+), BW", out.stdout)
        assert m and float(m.group(1)) > 0 and float(m.group(2)) > 0, out.stdout
        assert ("EmbeddingBag FWD-BWD(+SGD)" in out.stdout) == ("--run-baseline" in extra)
