"""bench.py --gpus N without a launcher starts N ranks itself (CPU test: BENCH_LAUNCH_TEST=1 replaces the GPU work by a
report, the ranks still form a gloo group), and refuses to print a one-GPU number for N GPUs it does not have."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, env=_env(BENCH_LAUNCH_TEST="1"), timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert [ln for ln in r.stdout.decode().split("\n") if ln.strip() and not ln.startswith("{")] == []     # stdout carries the JSON line and nothing else (gloo's chatter goes to stderr)
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.startswith("{")]
    assert len(lines) == 1                                      # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["max_rank_plus_1"] == 2.0     # both ranks joined the group


def test_gpus_8_without_eight_gpus_fails_instead_of_measuring_one():
    import torch
    if torch.cuda.device_count() >= 8:
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, env=_env(), timeout=300)
    assert r.returncode != 0 and b"n_gpus" not in r.stdout
    assert b"--gpus 8" in r.stderr


def test_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True,
                       env=_env(WORLD_SIZE="4", RANK="0", BENCH_LAUNCH_TEST="1"), timeout=300)
    assert r.returncode != 0
