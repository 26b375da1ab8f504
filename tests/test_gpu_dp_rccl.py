"""-m gpu: the data-parallel exchange on real RCCL. One GPU: a 1-rank "nccl" group (every collective is really
issued, hooks + side stream + arena ordering exercised; UNSLOTH_AMD_DP_FORCE=1), AND two ranks sharing that one GPU with
the collectives carried by gloo (`*_two_ranks_one_gpu_*`: world_size 2 x real GPU gradient sinks -- runs on every box).
Two or more visible GPUs: two ranks on RCCL, different batches, reduced LoRA gradients == the local replay of both batches
(skipped on a single-GPU box -- the driver's 8-GPU SCALE run is where N > 1 on RCCL is measured)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra_env, worker="_dp_worker.py"):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        # one node, no fabric: keep RCCL's bootstrap off interface / InfiniBand probing (seen to take 100 s on one box)
        env.setdefault("NCCL_SOCKET_IFNAME", "lo")
        env.setdefault("NCCL_IB_DISABLE", "1")
        env.setdefault("GLOO_SOCKET_IFNAME", "lo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
    return outs


@pytest.mark.parametrize("gc", ["off", "unsloth"])
def test_forced_one_rank_rccl_group(gc):
    outs = _run(1, {"UNSLOTH_AMD_DP_FORCE": "1", "DP_TEST_GC": gc})
    assert "rank 0/1 ok" in outs[0]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs")
@pytest.mark.parametrize("gc,bucket", [("off", 1 << 16), ("unsloth", 1 << 30)])
def test_two_rank_rccl_allreduce_matches_local_replay(gc, bucket):
    outs = _run(2, {"DP_TEST_GC": gc, "DP_TEST_BUCKET": str(bucket)})
    assert "rank 0/2 ok" in outs[0] and "rank 1/2 ok" in outs[1]


def test_full_finetune_forced_one_rank_rccl_group():
    """BASELINE config 3's exchange on real RCCL (one forced rank): per-bucket in-place reduce-scatter launched from the
    gradient sinks, sharded AdamW, in-place all-gather -- bit-identical to the same steps without any collective."""
    outs = _run(1, {"UNSLOTH_AMD_DP_FORCE": "1"}, worker="_fullft_worker.py")
    assert "rank 0/1 ok" in outs[0]


@pytest.mark.parametrize("gc,bucket", [("off", 1 << 16), ("unsloth", 1 << 30)])
def test_two_ranks_one_gpu_real_sinks_lora_arena(gc, bucket):
    """world_size 2 x REAL GPU gradient sinks on ONE device (collectives over gloo: RCCL refuses two ranks per device):
    one exchange per bucket per step, complete when launched, reduced gradients == single-process replay of both batches,
    also after zero_grad(set_to_none=True) and across a no_sync() accumulation; replicas identical after FlatAdamW."""
    outs = _run(2, {"DP_TEST_GC": gc, "DP_TEST_BUCKET": str(bucket), "DP_TEST_BACKEND": "gloo", "DP_TEST_ONE_DEVICE": "1"})
    assert "rank 0/2 ok [gloo" in outs[0] and "rank 1/2 ok [gloo" in outs[1], outs


def test_two_ranks_one_gpu_real_sinks_full_finetune():
    """BASELINE config 3's exchange with two ranks on one device: FullGradBuckets' sinks + sharded AdamW + all-gather;
    exchanged gradients == replay of both batches, replicas identical after three steps."""
    outs = _run(2, {"DP_TEST_BACKEND": "gloo", "DP_TEST_ONE_DEVICE": "1"}, worker="_fullft_worker.py")
    assert "rank 0/2 ok [gloo" in outs[0] and "rank 1/2 ok [gloo" in outs[1], outs
    assert "exchanged gradients == replay of 2 batches" in outs[0]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs")
def test_full_finetune_two_ranks_end_identical():
    outs = _run(2, {}, worker="_fullft_worker.py")
    assert "rank 0/2 ok" in outs[0] and "rank 1/2 ok" in outs[1]
