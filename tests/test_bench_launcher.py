"""`python bench.py --gpus N` must start N ranks by itself: the driver's scaling run has the command shape
of its 1-GPU run, and a bench that quietly times one GPU under `--gpus 8` would void the only multi-GPU evidence there is.
CPU, no GPU work: BENCH_DRY_LAUNCH=1 keeps the launcher, the torch.distributed.run rendezvous on 127.0.0.1 and the
one-JSON-line-from-rank-0 contract, and swaps RCCL for gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_launches_two_ranks_and_rank0_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"BENCH_DRY_LAUNCH": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # exactly one record, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dry_launch"] is True
    assert "launching 2 ranks" in r.stderr


def test_gpus_1_stays_in_process():
    r = _run(["--gpus", "1"], {"BENCH_DRY_LAUNCH": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1
    assert "launching" not in r.stderr


def test_fewer_devices_than_asked_is_an_error_not_a_one_gpu_run():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        return
    r = _run(["--gpus", "64"])
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "BENCH_DRY_LAUNCH": "1"}, drop=())
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "WORLD_SIZE=2" in r.stderr
