"""Which gradient arrival launches which bucket's reduce-scatter (full fine-tuning, 1 forced RCCL rank, 2 layers)?"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(UNSLOTH_AMD_DP_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0",
                  NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
from transformers import LlamaConfig  # noqa: E402

from unsloth_amd import FastLanguageModel  # noqa: E402
from unsloth_amd.full_finetune import FullGradBuckets, ShardedAdamW, full_finetune_step  # noqa: E402

cfg = LlamaConfig(hidden_size=1024, intermediate_size=2048, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
                  head_dim=128, vocab_size=4096, max_position_embeddings=2048, tie_word_embeddings=False)
model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=1024, dtype=torch.bfloat16, full_finetuning=True, device=dev,
                                             use_gradient_checkpointing=False)
opt = ShardedAdamW(model, lr=1e-5)
B = opt.buckets
name_of = {id(p): n for b in B.buckets for n, p in zip(b["names"], b["params"])}
log = []
real_count, real_launch = B._count, B._launch


def count(p):
    import inspect
    bi, _ = B._where[id(p)]
    fr = inspect.stack()
    who = " <- ".join(f"{f.function}:{f.lineno}" for f in fr[1:5])
    log.append(("arrive", bi, name_of[id(p)], B.buckets[bi]["pending"] + 1, B.buckets[bi]["expected"], who))
    return real_count(p)


def launch(bi):
    log.append(("LAUNCH", bi, "", B.buckets[bi]["pending"], B.buckets[bi]["expected"]))
    return real_launch(bi)


B._count, B._launch = count, launch
ids = torch.randint(0, 4096, (2, 512), device=dev)
batch = dict(input_ids=ids, labels=ids.clone())
for step in range(1):
    log.clear()
    full_finetune_step(model, batch, opt)
    torch.cuda.synchronize()
    print(f"--- step {step}: {sum(1 for l in log if l[0] == 'LAUNCH')} launches for {len(B.buckets)} buckets")
    for l in log:
        print("  ", *l)
dist.destroy_process_group()
