#!/usr/bin/env python
"""bench.py -- train tokens/sec + peak VRAM, Llama-3-8B QLoRA (NF4, r=16, all 7 projections), seq 2048,
bf16, on N MI355X of one node (BASELINE.json metric, configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = forward + backward + LoRA-grad exchange + AdamW step over one synthetic micro-batch per GPU
(weak scaling: per-GPU work fixed). Weights are random-init at the Llama-3-8B architecture and quantised to
bitsandbytes-format NF4 by our own quantiser; token ids ~ U[0, V), labels = ids, position_ids = arange
(int32, exercising the indexed RoPE path). Nothing is skipped inside the timed region.

The primary number runs under the API default `use_gradient_checkpointing="unsloth"` (the least-recompute schedule that
fits the free HBM: on an idle 288 GB part every layer keeps everything) with every NF4 weight decoded at every use INSIDE the
timed step (no decoded mirrors: they are opt-in, reported beside it as `value_with_resident_mirrors`). `vram_batch1_unsloth_min`
is the low-VRAM operating point (1 x 2048 tokens, layer inputs only); `gpu_baseline` is stock HuggingFace bf16 + torch LoRA +
SDPA timed on the same GPU on the same batch after everything else (`vs_gpu_baseline` = the ratios). `alt` holds the other operating points of the
reference: the checkpointing modes, batch 1 / 2, the PADDING-FREE PACKED step its SFT trainer runs by default
(trainer.py:903-912, utils/packing.py:241-284), BASELINE config 3 (full fine-tuning), config 4 (Qwen2-VL-7B, one image in
4096 tokens) and config 5 (Mistral-7B seq 4096: CE leg and the chunked GRPO log-prob leg), the DP path forced onto one
rank, decode. `--only NAME` times ONE of them as the whole run (what the per-point rocprofv3 kernel stats under
profiles/ are collected with).

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline     : the dominant kernel (the MFMA GEMM), ALGORITHMIC flops of its launches / their HIP-event
                 durations measured live in the timed region, against the 2.5 PFLOP/s dense bf16 peak
  cpu_baseline : the reference's torch-fp32 CPU composition timed on this box's host cores (bounded sample)
"""
import argparse
import json
import os
import sys
import time

# before the HIP / HSA runtime is initialised (the first torch.cuda call): the host driver only supports dmabuf IPC, and RCCL's
# intra-node transport fails with `hipIpcGetMemHandle: invalid argument` without it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GC_MODE = {"on": True, "off": False, "unsloth": "unsloth", "unsloth:min": "unsloth:min", "unsloth:all": "unsloth:all",
           "unsloth:auto": "unsloth:auto", "unsloth:attn": "unsloth:attn"}
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md (never the 2:1-sparse figure)
HBM_PEAK_GBPS = 8000.0
ONLY_POINTS = ("primary", "packed", "config1", "config4", "config5_ce", "config5_logprob", "dp_force", "batch1", "fullft")


def llama3_8b_config(n_layers=32, vocab=128256):
    from transformers import LlamaConfig
    return LlamaConfig(
        hidden_size=4096, intermediate_size=14336, num_hidden_layers=n_layers, num_attention_heads=32,
        num_key_value_heads=8, head_dim=128, vocab_size=vocab, rms_norm_eps=1e-5, max_position_embeddings=8192,
        rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, tie_word_embeddings=False,
        attention_bias=False, mlp_bias=False)


def tinyllama_1b_config(n_layers=22):
    """BASELINE config 1: TinyLlama-1.1B widths (32 query heads on 4 KV heads, head_dim 64)."""
    from transformers import LlamaConfig
    return LlamaConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=n_layers, num_attention_heads=32,
                       num_key_value_heads=4, head_dim=64, vocab_size=32000, rms_norm_eps=1e-5, max_position_embeddings=2048,
                       rope_parameters={"rope_type": "default", "rope_theta": 1e4}, tie_word_embeddings=False,
                       attention_bias=False, mlp_bias=False)


def mistral_7b_config(n_layers=32):
    """BASELINE config 5: Mistral-7B-v0.1 widths (sliding window 4096 == the sequence length: inactive, mistral.py:116-120)."""
    from transformers import MistralConfig
    return MistralConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=n_layers, num_attention_heads=32,
                         num_key_value_heads=8, head_dim=128, vocab_size=32000, rms_norm_eps=1e-5,
                         max_position_embeddings=32768, sliding_window=4096,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e4}, tie_word_embeddings=False)


def qwen2_vl_7b_config(n_layers=28, vit_depth=32):
    """BASELINE config 4: Qwen2-VL-7B (language 3584 / 18944 / 28:4 heads / vocab 152064 with q/k/v bias; ViT 1280 / 16 heads,
    patch 14, merge 2)."""
    from transformers import Qwen2VLConfig
    return Qwen2VLConfig(
        text_config=dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=n_layers, num_attention_heads=28,
                         num_key_value_heads=4, vocab_size=152064, max_position_embeddings=32768, rms_norm_eps=1e-6,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
                         tie_word_embeddings=False),
        vision_config=dict(depth=vit_depth, embed_dim=1280, hidden_size=3584, num_heads=16, mlp_ratio=4, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653)


KERNEL_OF = {"uamd_gemm_nt_256": "gemm_nt256_kernel<bf16>", "uamd_gemm_nn_256": "gemm_nt256_kernel<bf16>", "uamd_gemm_nt": "gemm_nt_kernel<bf16,dense>",
             "uamd_gemm_nt_nf4": "gemm_nt_kernel<bf16,NF4>"}


class GemmTimer:
    """HIP-event pairs around every MFMA GEMM launch and every flash-attention launch, recorded on the stream the kernel
    is launched on (torch's current stream == the stream passed through the C ABI). One record per launch:
    (start, end, algorithmic flops, algorithmic bytes)."""

    def __init__(self):
        from unsloth_amd.kernels import attention as A
        from unsloth_amd.kernels import utils as U
        self.U, self.A = U, A
        self.orig = U._launch_gemm
        self.orig_fwd, self.orig_bwd = A.attn_forward, A.attn_backward
        self.records = {}
        self.enabled = False
        self.steps_sampled = 0         # timed steps whose launches carry event pairs (bench --roofline-every)

    def install(self):
        U, A, orig, recs = self.U, self.A, self.orig, self.records

        def timed(X2d, groups, nf4, accumulate=False, nn=False):
            if not self.enabled:
                return orig(X2d, groups, nf4, accumulate, nn)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            flops = 2.0 * X2d.shape[0] * X2d.shape[1] * sum(g.N for g in groups)
            flops += sum(2.0 * X2d.shape[0] * g.R * g.N for g in groups if g.lora_xa)
            # algorithmic bytes: A once, every B once, every C written once (+ read once when accumulating)
            M_, K_ = X2d.shape
            nbytes = 2.0 * M_ * K_ + sum(2.0 * g.N * K_ + 2.0 * M_ * g.N * (2 if accumulate else 1) for g in groups)
            s.record()
            name = orig(X2d, groups, nf4, accumulate, nn)
            e.record()
            recs.setdefault(KERNEL_OF[name], []).append((s, e, flops, nbytes))
            return name

        pair_cache = {}

        def pairs(q, band, causal):
            """(query, key) pairs inside the mask, summed over the batch. Causal: T (T + 1) / 2 per row, sum_q (q - lo[q] + 1)
            under a packed / windowed band. Non-causal (the ViT): T * T per row, sum_q (hi[q] - lo[q] + 1) under a document
            band. A 0-d tensor (no host sync here) or a float."""
            B, T = q.shape[0], q.shape[1]
            if band is None:
                return B * T * (T + 1) / 2.0 if causal else float(B) * T * T
            lo, hi = band[0], band[1]
            key = (lo.data_ptr(), hi.data_ptr(), B, T, causal)
            if key not in pair_cache:
                last = torch.arange(T, device=lo.device, dtype=torch.int64).unsqueeze(0) if causal else hi.to(torch.int64)
                pair_cache[key] = (last - lo.to(torch.int64) + 1).sum()
            return pair_cache[key]

        def attn_timed(which, mult, fn):
            names = ("q", "k", "v", "scale", "band", "causal") if which == "fwd" else \
                    ("do", "q", "k", "v", "o", "lse", "scale", "band", "causal")

            def run(*args, **kw):
                if not self.enabled:
                    return fn(*args, **kw)
                bound = dict(zip(names, args), **kw)            # by position AND by keyword (causal=False calls)
                q, band, causal = bound["q"], bound.get("band"), bound.get("causal", True)
                Hq, D = q.shape[2], q.shape[3]
                unit = mult * 4.0 * D * Hq            # forward: S = Q K^T and O = P V; backward 2.5x (S again + 4 products)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = fn(*args, **kw)
                e.record()
                recs.setdefault("attention_" + which, []).append((s, e, (unit, pairs(q, band, causal)), 0.0))
                return out
            return run

        U._launch_gemm = timed
        A.attn_forward = attn_timed("fwd", 1.0, self.orig_fwd)
        A.attn_backward = attn_timed("bwd", 2.5, self.orig_bwd)

    def reset(self):
        self.records.clear()
        self.steps_sampled = 0

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            if not recs:
                continue
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            fl = sum((r[2][0] * float(r[2][1])) if isinstance(r[2], tuple) else r[2] for r in recs)
            out[name] = dict(launches=len(recs), steps_sampled=max(1, self.steps_sampled), total_ms=ms,
                             avg_us=ms * 1e3 / len(recs), flops=fl,
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                             alg_bytes=sum(r[3] for r in recs) / len(recs))
        return out


def fractions(gs, ms_per_step):
    """GEMM / attention roofline fractions of one operating point from a GemmTimer.summary()."""
    gem = {k: v for k, v in gs.items() if not k.startswith("attention_")}
    dom = max(gem.values(), key=lambda r: r["total_ms"]) if gem else None
    out = {"gemm_tflops": round(dom["tflops"], 1) if dom else None,
           "gemm_frac_of_mfma_peak": round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4) if dom else None,
           "gemm_share_of_step": round(dom["total_ms"] / dom["steps_sampled"] / ms_per_step, 3) if dom else None}
    af, ab = gs.get("attention_fwd"), gs.get("attention_bwd")
    if af:
        out["attention_fwd_frac_of_mfma_peak"] = round(af["tflops"] / MFMA_PEAK_TFLOPS, 4)
        out["attention_fwd_avg_us"] = round(af["avg_us"], 1)
    if ab:
        out["attention_bwd_frac_of_mfma_peak"] = round(ab["tflops"] / MFMA_PEAK_TFLOPS, 4)
        out["attention_bwd_avg_us"] = round(ab["avg_us"], 1)
    if af and ab:
        ms = af["total_ms"] + ab["total_ms"]
        out["attention_frac_of_mfma_peak"] = round((af["flops"] + ab["flops"]) / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
        out["attention_share_of_step"] = round(ms / af["steps_sampled"] / ms_per_step, 3)
    return out


class TorchLoRALinear(torch.nn.Module):
    """What PEFT's lora.Linear computes on a frozen nn.Linear (peft is not installed here): base(x) + B(A(x)) * alpha / r, the
    factors fp32 parameters used in the activation dtype under autocast."""

    def __init__(self, base, r, alpha, gen):
        super().__init__()
        self.base, self.scale = base, alpha / r
        dev = base.weight.device
        self.lora_A = torch.nn.Parameter((torch.rand(r, base.in_features, generator=gen) * 2 - 1).mul_(base.in_features ** -0.5).to(dev))
        self.lora_B = torch.nn.Parameter((torch.randn(base.out_features, r, generator=gen) * 0.02).to(dev))

    def forward(self, x):
        F_ = torch.nn.functional
        return self.base(x) + F_.linear(F_.linear(x, self.lora_A.to(x.dtype)), self.lora_B.to(x.dtype)) * self.scale


def hf_gpu_baseline(cfg, dev, B, T, r, steps=3, warmup=2, seed=0, checkpointing=False):
    """Stock HuggingFace `LlamaForCausalLM` (its own RMSNorm / RoPE / SwiGLU / loss modules, attn_implementation="sdpa") in
    bf16 with torch LoRA r on the same 7 projections, bf16 autocast, torch's fused AdamW on the factors, no gradient
    checkpointing -- the same B x T synthetic batch, forward + backward + optimizer step, on this GPU. Base weights are
    bf16, not NF4: bitsandbytes is not installed here, so HF cannot run a 4-bit model at all; that spares the baseline the
    dequantisation work and costs it ~10 GB of weights. Every class-level patch of unsloth_amd is undone first."""
    from transformers import AutoModelForCausalLM
    from unsloth_amd.kernels import unpatch_rms_layernorm
    from unsloth_amd.kernels.cross_entropy_loss import unpatch_loss_functions
    from unsloth_amd.models import llama as _L
    _L.unpatch_all()
    unpatch_rms_layernorm()
    unpatch_loss_functions()
    torch.manual_seed(3407)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            hf = AutoModelForCausalLM.from_config(cfg, attn_implementation="sdpa")
    finally:
        torch.set_default_dtype(old)
    hf.to(torch.bfloat16)
    for p in hf.parameters():
        p.requires_grad_(False)
    g_ = torch.Generator(device="cpu").manual_seed(3407)
    for layer in hf.model.layers:
        for parent, names in ((layer.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (layer.mlp, ("gate_proj", "up_proj", "down_proj"))):
            for n in names:
                setattr(parent, n, TorchLoRALinear(getattr(parent, n), r, r, g_))
    params = [p for p in hf.parameters() if p.requires_grad]
    o = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.01, fused=True)
    hf.train()
    hf.config.use_cache = False
    if checkpointing:           # HF's own per-layer activation checkpointing (non-reentrant: the frozen embeddings give no input gradient)
        hf.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})
    gi = torch.Generator(device="cpu").manual_seed(seed)
    bt = [torch.randint(0, cfg.vocab_size, (B, T), generator=gi).to(dev) for _ in range(2)]

    def step(i):
        ids = bt[i % 2]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = hf(input_ids=ids, labels=ids, use_cache=False).loss
        loss.backward()
        o.step()
        o.zero_grad(set_to_none=True)
        return loss.detach()

    try:
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        times, losses = [], []
        for i in range(steps):
            t0 = time.perf_counter()
            losses.append(step(i))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        med = sorted(times)[len(times) // 2]
        return {"value": round(B * T / med, 1), "unit": "tokens/s", "ms_per_step": round(med * 1e3, 2),
                "peak_vram_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2), "steps": steps, "timing": "median step",
                "what": "stock HF LlamaForCausalLM bf16 (sdpa) + torch LoRA r=%d on 7 projections + fused torch AdamW, bf16 autocast, "
                        "%s, %d x %d tokens, same GPU, after the runs above"
                        % (r, "HF gradient checkpointing (every layer, non-reentrant)" if checkpointing else "no gradient checkpointing", B, T),
                "base_weights": "bf16 (HF cannot run NF4 here: no bitsandbytes)", "trainable_params": sum(p.numel() for p in params),
                "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)]}
    finally:
        del hf, o, params
        torch.cuda.empty_cache()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: become the launcher. One process per GPU under
    torch.distributed.run on 127.0.0.1 (the driver's own command shape); the ranks find RANK / LOCAL_RANK / WORLD_SIZE in
    their environment and rank 0 prints the one JSON line on the stdout this process was given. Fewer than N visible
    devices is an error (exit 2), never a quiet 1-GPU run. BENCH_DRY_LAUNCH=1 (CPU test of this path): no device check,
    the ranks rendezvous over gloo and report who took part."""
    import socket
    dry = os.environ.get("BENCH_DRY_LAUNCH", "0") == "1"
    if not dry:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write(f"[bench] --gpus {n} but {have} GPU(s) visible: refusing to report a {n}-GPU number\n")
            raise SystemExit(2)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks ({' '.join(cmd[1:8])} ...)\n")
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def dry_launch(a, real_stdout):
    """BENCH_DRY_LAUNCH=1: the launcher / rendezvous / one-line-from-rank-0 path without a GPU (tests/test_bench_launcher.py)."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    took_part = 1
    if world > 1:
        dist.init_process_group("gloo")
        one = torch.ones(1)
        dist.all_reduce(one)
        took_part = int(one.item())
    if rank == 0:
        rec = {"metric": "bench.py launcher dry run (no GPU work)", "value": 0.0, "unit": "tokens/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "rccl_ranks": took_part, "dry_launch": True}
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("BENCH_BATCH", 4)), help="sequences per GPU per step")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--gc", choices=sorted(GC_MODE), default=os.environ.get("BENCH_GC", "unsloth"),
                    help="gradient checkpointing for the primary number. unsloth (the API default of from_pretrained / "
                         "get_peft_model): the least-recompute schedule that fits the free HBM (models/fast_layer.py "
                         "auto_schedule); off: no checkpointing; on: torch's reentrant per-layer checkpoint (layer inputs only, "
                         "one extra forward per layer); unsloth:<policy>: a fixed selective-recompute policy")
    ap.add_argument("--alt-steps", type=int, default=int(os.environ.get("BENCH_ALT_STEPS", 5)),
                    help="also time this many steps at the other operating points (reported under 'alt'); 0 = skip")
    ap.add_argument("--only", choices=ONLY_POINTS, default=os.environ.get("BENCH_ONLY", "primary"),
                    help="time ONE operating point with --steps / --warmup as the whole run (per-point rocprofv3 profiles); "
                         "the JSON line then describes that point. 'primary' (default) = the BASELINE metric + every alt point")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-baseline", action="store_true", default=os.environ.get("BENCH_GPU_BASELINE", "1") == "0",
                    help="skip the stock-HuggingFace bf16 + torch-LoRA + SDPA step timed on the same GPU after everything else")
    ap.add_argument("--roofline-every", type=int, default=int(os.environ.get("BENCH_ROOFLINE_EVERY", 4)),
                    help="HIP-event pairs around the GEMM launches on every Nth timed step (the first one always). An event "
                         "pair drains the queue around its kernel: ~12 us per GEMM, 292 GEMMs per step = 1.4 %% of the step "
                         "when every step is instrumented (profiles/r03final_step_sequence.csv: all 3.5 ms of idle gaps of a "
                         "step sit before a GEMM or before the kernel that follows one). 1 = every step")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a.gpus, sys.argv[1:])               # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        # a scaling run that quietly measures another world size is worse than no run
        sys.stderr.write(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: pass the same N to torchrun and to --gpus\n")
        raise SystemExit(2)
    # stdout must carry exactly ONE line, the JSON record. Libraries print to the C-level stdout behind Python's back
    # (RCCL's version banner at the first collective, flushed at exit, i.e. AFTER the record): keep the real stdout
    # aside for the record and send everything else written to fd 1 to stderr.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if os.environ.get("BENCH_DRY_LAUNCH", "0") == "1":
        return dry_launch(a, real_stdout)
    if world > 1 and os.environ.get("BENCH_ALT_MULTI", "0") != "1":
        a.alt_steps = 0            # the other operating points are a 1-GPU report; a scaling run times the primary only
    if a.only != "primary":
        a.alt_steps = 0
        a.no_cpu_baseline = a.no_gpu_baseline = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda unavailable); the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if a.only == "dp_force":
        os.environ["UNSLOTH_AMD_DP_FORCE"] = "1"
    force_dp = os.environ.get("UNSLOTH_AMD_DP_FORCE", "0") == "1"     # 1-rank RCCL group: exercises the DP path on one GPU

    def init_rccl():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one node over xGMI: RCCL's bootstrap needs no NIC; keep it off interface / InfiniBand probing
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)          # "nccl" IS RCCL on ROCm
    if world > 1 or force_dp:
        init_rccl()

    from unsloth_amd import FastLanguageModel
    from unsloth_amd.dp import LoRAGradArena
    from unsloth_amd.trainer import make_optimizer, training_step

    timer = GemmTimer()
    timer.install()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def randomize_lora_b(model, seed=3407):
        g_ = torch.Generator(device="cpu").manual_seed(seed)
        n_train = 0
        for n, p in model.named_parameters():
            if p.requires_grad:
                n_train += p.numel()
                if "lora_B" in n:           # non-zero B so every LoRA gradient is exercised with real numbers
                    p.data.copy_((torch.randn(p.shape, generator=g_) * 0.02).to(p.device))
        return n_train

    last_timing = {}

    def timed_steps(step_fn, steps, warmup, per_step=False):
        """W untimed + exactly K timed calls of step_fn(i), barrier + synchronize on both sides, MAX over ranks.
        per_step (the short `alt` points only, never the primary): every step is bracketed on its own and the MEDIAN step
        time x K is returned -- a 5-step point (3 until round 6) is otherwise at the mercy of one allocator stall after empty_cache()."""
        losses = []
        for i in range(warmup):
            losses.append(step_fn(i))
        sync()
        torch.cuda.reset_peak_memory_stats()
        timer.reset()
        t0 = time.perf_counter()
        if per_step:
            times = []
            for i in range(steps):
                ts = time.perf_counter()
                timer.enabled = (i == 0)                # the MEDIAN step is then one without event pairs
                timer.steps_sampled += int(timer.enabled)
                losses.append(step_fn(i))
                sync()
                times.append(time.perf_counter() - ts)
            dt = sorted(times)[len(times) // 2] * steps
        else:
            every = max(1, a.roofline_every)
            for i in range(steps):
                timer.enabled = (i % every == 0)        # roofline sample: the launches of every Nth timed step
                timer.steps_sampled += int(timer.enabled)
                losses.append(step_fn(i))
            sync()
            dt = time.perf_counter() - t0
        timer.enabled = False
        peak = torch.cuda.max_memory_allocated()
        last_timing["per_rank_ms_per_step"] = [round(dt / steps * 1e3, 2)]
        if world > 1:
            # every rank's own clock over the same K steps (the line's ms_per_step is their MAX): a straggler shows here
            mine = torch.tensor([dt / steps * 1e3], device=dev, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            last_timing["per_rank_ms_per_step"] = [round(float(t), 2) for t in allr]
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
            pk = torch.tensor([peak], device=dev, dtype=torch.int64)
            dist.all_reduce(pk, op=dist.ReduceOp.MAX)
            peak = int(pk)
        return dt, peak, [float(l) for l in losses], timer.summary()

    def point(tokens_per_step, dt, steps, peak, gs, **extra):
        ms = dt / steps * 1e3
        rec = {"steps": steps, "value": round(tokens_per_step * steps * world / dt, 1), "unit": "tokens/s",
               "ms_per_step": round(ms, 2), "peak_vram_gb": round(peak / 2**30, 2)}
        rec.update(fractions(gs, ms))
        rec.update(extra)
        return rec

    # ------------------------------------------------------------------------------------------------------------
    # the Llama-3-8B QLoRA model of the BASELINE metric (primary + the points that share it)
    cfg = llama3_8b_config(a.layers)
    B, T, V = a.batch, a.seq, cfg.vocab_size
    gi = torch.Generator(device="cpu").manual_seed(rank)       # different data per rank
    need_llama = a.only in ("primary", "packed", "dp_force", "batch1")
    model = opt = arena = None
    n_train = 0
    setup_s = 0.0
    if need_llama:
        t_setup = time.time()
        model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=max(a.seq, 8192), dtype=torch.bfloat16,
                                                     load_in_4bit=True, device=dev, random_state=3407,
                                                     use_gradient_checkpointing=GC_MODE[a.gc])
        # NOTE: for_training() below re-applies the checkpointing mode per measurement
        model = FastLanguageModel.get_peft_model(model, r=a.rank, lora_alpha=a.rank, lora_dropout=0.0, bias="none",
                                                 use_gradient_checkpointing=GC_MODE[a.gc], random_state=3407)
        n_train = randomize_lora_b(model)
        arena = LoRAGradArena(model) if (world > 1 or force_dp) else None
        # optim.FlatAdamW: parameters / gradients / moments in flat arenas, one launch per step (the gradients live in the
        # data-parallel arena when there is one; UNSLOTH_AMD_FLAT_ADAMW=0 = torch's fused AdamW)
        opt = make_optimizer(model, lr=2e-4, arena=arena)
        setup_s = time.time() - t_setup

    def make_batches(bs, seq=None):
        seq = seq or T
        out = []
        for _ in range(2):
            ids = torch.randint(0, V, (bs, seq), generator=gi).to(dev)
            pos = torch.arange(seq, dtype=torch.int32, device=dev).unsqueeze(0).expand(bs, seq).contiguous()
            out.append(dict(input_ids=ids, labels=ids.clone(), position_ids=pos))
        return out, torch.tensor((seq - 1) * bs * world, device=dev)      # global non-ignored targets per step

    def make_packed_batches(total, vocab):
        """The reference's default SFT batch (trainer.py:903-912: padding_free): documents of mixed length concatenated
        into ONE row, `packed_seq_lengths` + position ids restarting per document (utils/packing.py:241-284), the first
        token of every document unlabelled. Lengths ~ a chat-SFT mix: 64 ... 2048 tokens, mean ~ 640."""
        from unsloth_amd.utils.packing import enable_padding_free_metadata
        out = []
        for _ in range(2):
            lens, left = [], total
            while left > 0:
                n = int(torch.randint(64, 2049, (1,), generator=gi))
                n = min(n if int(torch.randint(0, 3, (1,), generator=gi)) == 0 else max(64, n // 4), left)
                lens.append(n)
                left -= n
            docs = [torch.randint(0, vocab, (n,), generator=gi).tolist() for n in lens]
            out.append(enable_padding_free_metadata(docs, device=dev))
        n_items = sum(int((b["labels"][:, 1:] != -100).sum()) for b in out) // 2
        return out, torch.tensor(n_items * world, device=dev), [len(b["packed_seq_lengths"]) for b in out]

    def measure(gc_mode, steps, warmup, data=None, per_step=False):
        bt, ni = data if data is not None else (batches, n_items)
        model.for_training(use_gradient_checkpointing=gc_mode)
        return timed_steps(lambda i: training_step(model, bt[i % 2], opt, arena, ni), steps, warmup, per_step)

    # ------------------------------------------------------------------------------------------------------------
    # operating points that build their own model (BASELINE configs 3, 4, 5)
    def run_config1(rows, steps, warmup, per_step):
        """BASELINE config 1 ON THE GPU (the configuration itself is the reference's CPU plumbing case; its CPU timing is
        cpu_baseline.config1_tinyllama_direct): TinyLlama-1.1B widths, 16-bit base, LoRA r=8 on the 7 projections, seq 512,
        `rows` rows per step (1 = the configuration as stated: launch-bound; 16 = 8192 tokens, the headline's step size).
        head_dim 64 on the hand attention kernels without padded copies (round 6)."""
        tcfg = tinyllama_1b_config(22 if a.layers == 32 else a.layers)
        m, _ = FastLanguageModel.from_pretrained(config=tcfg, max_seq_length=2048, dtype=torch.bfloat16, load_in_4bit=False,
                                                 device=dev, random_state=3407, use_gradient_checkpointing=GC_MODE[a.gc])
        m = FastLanguageModel.get_peft_model(m, r=8, lora_alpha=8, use_gradient_checkpointing=GC_MODE[a.gc], random_state=3407)
        randomize_lora_b(m)
        o = make_optimizer(m, lr=2e-4)
        m.for_training(use_gradient_checkpointing=GC_MODE[a.gc])
        S = 512
        try:
            bt = []
            for _ in range(2):
                ids = torch.randint(0, 32000, (rows, S), generator=gi).to(dev)
                pos = torch.arange(S, dtype=torch.int32, device=dev).unsqueeze(0).expand(rows, S).contiguous()
                bt.append(dict(input_ids=ids, labels=ids.clone(), position_ids=pos))
            ni = torch.tensor((S - 1) * rows, device=dev)
            dt, peak, losses, gs = timed_steps(lambda i: training_step(m, bt[i % 2], o, None, ni), steps, warmup, per_step)
            return point(rows * S, dt, steps, peak, gs, rows=rows, seq_len=S, lora_rank=8, base="bf16 (no NF4)", head_dim=64,
                         loss_first_last=[round(losses[0], 4), round(losses[-1], 4)])
        finally:
            if hasattr(o, "close"):
                o.close()
            del m, o
            torch.cuda.empty_cache()

    def run_config5(leg, steps, warmup, per_step):
        """BASELINE config 5: Mistral-7B widths, NF4 + LoRA r=16, seq 4096, 2 rows per GPU. `ce`: the SFT / DPO-style CE step;
        `logprob`: one GRPO policy step -- [left-padded prompt | right-padded completion] rows packed into one varlen forward
        that returns hidden states, lm_head + log-softmax on the completion positions only in row chunks
        (models/rl_replacements.py; ref rl_replacements.py:1517-1700), the clipped objective, backward, AdamW."""
        from unsloth_amd.models.rl_replacements import grpo_accumulated_loss
        mcfg = mistral_7b_config(a.layers)
        m, _ = FastLanguageModel.from_pretrained(config=mcfg, max_seq_length=8192, dtype=torch.bfloat16, load_in_4bit=True,
                                                 device=dev, random_state=3407, use_gradient_checkpointing=GC_MODE[a.gc])
        m = FastLanguageModel.get_peft_model(m, r=16, lora_alpha=16, use_gradient_checkpointing=GC_MODE[a.gc], random_state=3407)
        randomize_lora_b(m)
        o = make_optimizer(m, lr=2e-4)
        m.for_training(use_gradient_checkpointing=GC_MODE[a.gc])
        rows, S = 2, 4096
        try:
            if leg == "ce":
                bt = []
                for _ in range(2):
                    ids = torch.randint(0, 32000, (rows, S), generator=gi).to(dev)
                    pos = torch.arange(S, dtype=torch.int32, device=dev).unsqueeze(0).expand(rows, S).contiguous()
                    bt.append(dict(input_ids=ids, labels=ids.clone(), position_ids=pos))
                ni = torch.tensor((S - 1) * rows, device=dev)
                dt, peak, losses, gs = timed_steps(lambda i: training_step(m, bt[i % 2], o, None, ni), steps, warmup, per_step)
                return point(rows * S, dt, steps, peak, gs, rows=rows, seq_len=S, lora_rank=16,
                             loss_first_last=[round(losses[0], 4), round(losses[-1], 4)])
            prompt, comp = 1024, 3072
            data = []
            for _ in range(2):
                ids = torch.randint(0, 32000, (rows, S), generator=gi)
                mask = torch.ones(rows, S, dtype=torch.int64)
                p_len = torch.randint(prompt // 2, prompt + 1, (rows,), generator=gi)
                c_len = torch.randint(comp // 2, comp + 1, (rows,), generator=gi)
                for r_ in range(rows):
                    mask[r_, :prompt - int(p_len[r_])] = 0                 # prompts are left-padded
                    mask[r_, prompt + int(c_len[r_]):] = 0                 # completions right-padded
                cmask = mask[:, prompt:].clone()
                adv = torch.randn(rows, generator=gi)
                old = -torch.rand(rows, comp, generator=gi) * 3.0
                data.append(tuple(x.to(dev) for x in (ids, mask, cmask, adv, old)))
            toks = sum(int(d[1].sum()) for d in data) / 2.0

            def grpo_step(i):
                ids, mask, cmask, adv, old = data[i % 2]
                loss = grpo_accumulated_loss(m, ids, mask, comp, cmask, adv, old_logps=old, beta=0.0)[0]
                loss.backward()
                o.step()
                o.zero_grad()
                return loss.detach()
            dt, peak, losses, gs = timed_steps(grpo_step, steps, warmup, per_step)
            return point(toks, dt, steps, peak, gs, rows=rows, seq_len=S, lora_rank=16, logprob_chunks=4,
                         tokens_counted="non-padding tokens of the packed forward (mean of the two batches): %d" % toks,
                         completion_logprobs_per_step=int(sum(int(d[2].sum()) for d in data) / 2),
                         objective_first_last=[round(losses[0], 5), round(losses[-1], 5)])
        finally:
            if hasattr(o, "close"):
                o.close()
            del m, o
            torch.cuda.empty_cache()

    def run_config4(steps, warmup, per_step):
        """BASELINE config 4: FastVisionModel at Qwen2-VL-7B's widths (ViT depth 32 / 1280 wide; language tower 28 layers NF4),
        LoRA r=32 on both towers, ONE 896 x 896 image (4096 patches -> 1024 merged tokens) inside a 4096-token row:
        pixel_values -> patch-embed -> ViT -> merger -> scatter -> mrope positions -> fused tower -> loss -> backward -> AdamW."""
        from unsloth_amd import FastVisionModel
        vcfg = qwen2_vl_7b_config(28 if a.layers == 32 else a.layers, 32 if a.layers == 32 else 2)
        m, _ = FastVisionModel.from_pretrained(config=vcfg, max_seq_length=4096, load_in_4bit=True, device=dev,
                                               use_gradient_checkpointing=GC_MODE[a.gc])
        m = FastVisionModel.get_peft_model(m, r=32, lora_alpha=32, use_gradient_checkpointing=GC_MODE[a.gc])
        randomize_lora_b(m)
        o = make_optimizer(m, lr=2e-4)
        m.for_training(use_gradient_checkpointing=GC_MODE[a.gc])
        grid, S, pre = (1, 64, 64), 4096, 700
        n_img = grid[0] * (grid[1] // 2) * (grid[2] // 2)
        bt = []
        for _ in range(2):
            row = torch.cat([torch.randint(0, 150000, (pre,), generator=gi), torch.tensor([vcfg.vision_start_token_id]),
                             torch.full((n_img,), vcfg.image_token_id), torch.tensor([vcfg.vision_end_token_id]),
                             torch.randint(0, 150000, (S - pre - n_img - 2,), generator=gi)])
            ids = row.unsqueeze(0)
            labels = ids.clone()
            labels[ids == vcfg.image_token_id] = -100
            pix = torch.randn(grid[0] * grid[1] * grid[2], 3 * 2 * 14 * 14, generator=gi)
            bt.append({k: v.to(dev) for k, v in dict(input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels,
                                                     pixel_values=pix, image_grid_thw=torch.tensor([grid])).items()})
        ni = torch.tensor(int((bt[0]["labels"][:, 1:] != -100).sum()), device=dev)
        try:
            dt, peak, losses, gs = timed_steps(lambda i: training_step(m, bt[i % 2], o, None, ni), steps, warmup, per_step)
            return point(S, dt, steps, peak, gs, rows=1, seq_len=S, image="896x896 -> 4096 patches -> 1024 tokens", lora_rank=32,
                         vit="depth %d: HIP 2-D RoPE + non-causal flash attention + QuickGELU + LayerNorm, LoRA_W linears (models/vision_tower.py)" % vcfg.vision_config.depth,
                         trainable_params=sum(p.numel() for p in m.parameters() if p.requires_grad),
                         loss_first_last=[round(losses[0], 4), round(losses[-1], 4)])
        finally:
            if hasattr(o, "close"):
                o.close()
            del m, o
            torch.cuda.empty_cache()

    def run_fullft(steps, warmup):
        # BASELINE config 3 on ONE GPU: the same architecture fully trainable in bf16 (dense dW GEMMs, norm / lm_head /
        # embedding gradients, flat buckets, fp32-master AdamW: 16 + 16 + 96 GB of the 288), world size 1 = no collective
        from unsloth_amd.full_finetune import ShardedAdamW, full_finetune_step
        fmodel, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=a.seq, dtype=torch.bfloat16,
                                                      full_finetuning=True, device=dev, random_state=3407,
                                                      use_gradient_checkpointing=False)
        fopt = ShardedAdamW(fmodel, lr=1e-5)
        fb, fni = make_batches(B)
        try:
            dt, peak, fl, gs = timed_steps(lambda i: full_finetune_step(fmodel, fb[i % 2], fopt, fni), steps, warmup, True)
            n_all = sum(p.numel() for p in fmodel.parameters())
            return point(B * T, dt, steps, peak, gs, batch=B, timing="median step", trainable_params=n_all,
                         model_tflops_per_s=round(6.0 * n_all * B * T * steps / dt / 1e12, 1),
                         loss_first_last=[round(float(fl[0]), 4), round(float(fl[-1]), 4)])
        finally:
            fopt.buckets.close()
            del fmodel, fopt
            torch.cuda.empty_cache()
            os.environ["UNSLOTH_ENABLE_FULL_FINETUNING"] = "0"

    def guarded(alt, tag, fn):
        try:
            alt[tag] = fn()
        except Exception as ex:                     # an operating point must never take the primary record down with it
            alt[tag] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            torch.cuda.empty_cache()

    PACKED_TAG = "packed_padding_free_1x8192 (the reference's default SFT batch: mixed-length documents in one row)"
    C4_TAG = "config4_qwen2_vl_7b_nf4_r32_seq4096_one_image"
    C1_TAG = "config1_tinyllama_1.1b_lora_r8_seq512_on_the_gpu"
    C5CE_TAG = "config5_mistral_7b_nf4_r16_seq4096_ce_leg"
    C5LP_TAG = "config5_mistral_7b_nf4_r16_seq4096_grpo_logprob_leg"

    # ------------------------------------------------------------------------------------------------------------
    if a.only != "primary":
        # ONE operating point as the whole run (rocprofv3 kernel stats per point)
        if a.only == "packed":
            pdata, pni, ndocs = make_packed_batches(B * T, V)
            dt, peak, losses, gs = measure(GC_MODE[a.gc], a.steps, a.warmup, (pdata, pni))
            rec = point(B * T, dt, a.steps, peak, gs, documents_per_batch=ndocs, gradient_checkpointing=GC_MODE[a.gc])
        elif a.only in ("dp_force", "batch1"):
            bs = 1 if a.only == "batch1" else B
            data = make_batches(bs)
            dt, peak, losses, gs = measure(GC_MODE[a.gc], a.steps, a.warmup, data)
            rec = point(bs * T, dt, a.steps, peak, gs, batch=bs, gradient_checkpointing=GC_MODE[a.gc],
                        dp_buckets=len(arena.buckets) if arena is not None else None,
                        collectives_issued_per_step=(arena.collectives / (a.steps + a.warmup)) if arena is not None else None)
        elif a.only == "config4":
            rec = run_config4(a.steps, a.warmup, False)
        elif a.only == "config1":
            rec = run_config1(int(os.environ.get("BENCH_CONFIG1_ROWS", 16)), a.steps, a.warmup, False)
        elif a.only == "fullft":
            rec = run_fullft(a.steps, a.warmup)
        else:
            rec = run_config5("ce" if a.only == "config5_ce" else "logprob", a.steps, a.warmup, False)
        rec = dict({"metric": "operating point '%s' of bench.py (not the BASELINE headline)" % a.only, "n_gpus": world,
                    "warmup": a.warmup, "higher_is_better": True, "dtype": "bf16", "data": "synthetic"}, **rec)
        if rank == 0:
            os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    # the PRIMARY measurement first, on the freshly initialised model (loss_first_last starts at ~ln V); the other operating
    # points afterwards (their warm-up / timing steps keep training the same adapters, which no longer matters)
    batches, n_items = make_batches(B)
    if arena is not None:
        arena.timing = True              # events around the waits of finish(): what the overlap with the backward did not hide
    dt, peak, loss_vals, gs = measure(GC_MODE[a.gc], a.steps, a.warmup)
    primary_per_rank = list(last_timing.get("per_rank_ms_per_step", []))
    dp_diag = None
    if arena is not None:
        # (warm-up steps' waits are in the sum too: divide by all the steps that ran)
        exposed = arena.exposed_ms() / max(1, a.steps + a.warmup)
        arena.timing = False
        ex = torch.tensor([exposed], device=dev, dtype=torch.float64)
        exs = [torch.zeros_like(ex) for _ in range(world)]
        if world > 1:
            dist.all_gather(exs, ex)
        else:
            exs = [ex]
        dp_diag = dict(dp_exposed_ms=round(max(float(t) for t in exs), 3), dp_exposed_ms_per_rank=[round(float(t), 3) for t in exs],
                       dp_buckets=len(arena.buckets), dp_bucket_mb=[round((e - s_) * 4 / 2**20, 1) for s_, e, _ in arena.buckets],
                       dp_collectives_per_step=round(arena.collectives / max(1, a.steps + a.warmup), 2),
                       note="dp_exposed_ms = HIP events on the compute stream around the waits of LoRAGradArena.finish(), per step, "
                            "MAX over ranks: the part of the exchange the backward did not hide")
    base_ = model.get_base_model().model
    sched = getattr(base_, "_uamd_auto_policy", None)
    primary_policy = None
    if sched is not None:
        from unsloth_amd.models import fast_layer as _fl
        pol = sched[1]
        primary_policy = ("all (every layer keeps everything)" if pol == _fl.POLICIES["all"] else
                          "attn (every layer re-runs norm2 + gate/up)" if pol == _fl.POLICIES["attn"] else
                          "all*%d,attn" % pol[0][0])
    from unsloth_amd import nf4 as _nf4p
    primary_mirrors = bool(_nf4p.mirrors_on(base_) and _nf4p.resident_count(base_) > 0)
    primary_mirror_gb = round(_nf4p.resident_bytes(base_) / 2**30, 2)
    with_mirrors = batch1_min = gpu_base = None
    alt = None
    if a.alt_steps > 0:
        # the other checkpointing modes at the primary batch, then batch 1 and 2 without checkpointing
        alt = {}

        def alt_point(tag, gc_mode, bs, data=None, **extra):
            if data is None and bs != B:
                data = make_batches(bs)
            adt, apeak, _, ags = measure(gc_mode, a.alt_steps, 3, data, per_step=True)     # 3 warm-up steps: allocator growth after empty_cache(), decoded mirrors
            alt[tag] = point(bs * T, adt, a.alt_steps, apeak, ags, gradient_checkpointing=gc_mode, batch=bs,
                             timing="median step x steps", **extra)
            del data
            torch.cuda.empty_cache()
        for tag, mode in (("gc_torch_reentrant (reference's True)", True),
                          ("gc_unsloth (API default: least-recompute schedule that fits the free HBM)", "unsloth"),
                          ("gc_unsloth_attn (fixed: keep attention block, re-run gate/up)", "unsloth:attn"),
                          ("gc_unsloth_min (keep layer inputs only)", "unsloth:min"), ("gc_off", False)):
            if mode != GC_MODE[a.gc]:
                alt_point(tag, mode, B)
        # the default spelling on a CROWDED GPU: 10.5 GB of HBM left for the step beyond the resident weights
        os.environ["UNSLOTH_AMD_GC_FREE_GB"] = "10.5"
        base_._uamd_auto_policy = None
        alt_point("gc_unsloth_with_10.5_GB_free (UNSLOTH_AMD_GC_FREE_GB=10.5: the same spelling falls back to 'attn')", "unsloth", B)
        del os.environ["UNSLOTH_AMD_GC_FREE_GB"]
        base_._uamd_auto_policy = None
        for bs in (1, 2):
            if bs != B:
                alt_point(f"batch_{bs}_gc_off", False, bs)
        # the VRAM half of the metric at the reference's own operating point: ONE 2048-token row, only the layer inputs kept
        # (SURVEY 9.11 budgets 7-8 GB for it). Everything that lives in HBM counts: NF4 base, bf16 embeddings + lm_head, LoRA
        # factors / gradients / AdamW moments, decode scratch, activations.
        B1MIN_TAG = "batch_1_gc_unsloth_min (1 x 2048 tokens, layer inputs only: the low-VRAM operating point)"
        alt_point(B1MIN_TAG, "unsloth:min", 1)
        batch1_min = {k: alt[B1MIN_TAG][k] for k in ("value", "ms_per_step", "peak_vram_gb")}
        batch1_min.update(batch=1, seq_len=T, gradient_checkpointing="unsloth:min", survey_budget_gb="7-8 (SURVEY 9.11)")
        # the reference's DEFAULT SFT step: padding-free packed row (band attention + indexed RoPE with restarting positions)
        pdata, pni, ndocs = make_packed_batches(B * T, V)
        alt_point(PACKED_TAG, GC_MODE[a.gc], 1, (pdata, pni), documents_per_batch=ndocs, tokens_per_row=B * T)
        alt[PACKED_TAG]["value"] = round(alt[PACKED_TAG]["value"] * B, 1)         # alt_point counted bs * T tokens with bs = 1
        alt[PACKED_TAG]["batch"] = "1 x %d" % (B * T)
        del pdata
        if os.environ.get("BENCH_RESIDENT_ALT", "1") == "1":
            # decoded bf16 mirrors of the NF4 weights (+2 B per projection parameter, no decode launches, bit-identical steps):
            # OPT-IN since round 5 (UNSLOTH_AMD_RESIDENT_WEIGHTS=auto | 1, nf4.set_resident) -- the headline above decodes NF4
            # inside every step, as the reference does. Here: this model's projections with mirrors, at the primary batch and at 1.
            from unsloth_amd import nf4 as _nf4
            _nf4.set_resident(True, model=base_)
            MIRROR_TAG = "gc_unsloth_with_decoded_weight_mirrors (opt-in UNSLOTH_AMD_RESIDENT_WEIGHTS=auto | 1: no NF4 decode in the step)"
            alt_point(MIRROR_TAG, "unsloth", B)
            with_mirrors = {k: alt[MIRROR_TAG][k] for k in ("value", "ms_per_step", "peak_vram_gb", "steps", "timing")}
            with_mirrors["mirror_gb"] = round(_nf4.resident_bytes(base_) / 2**30, 2)
            alt_point("batch_1_gc_off_with_decoded_weight_mirrors (UNSLOTH_AMD_RESIDENT_WEIGHTS=1)", False, 1)
            _nf4.set_resident(False, model=base_)
            base_._uamd_auto_policy = None
            torch.cuda.empty_cache()
        if os.environ.get("BENCH_STEP_DECODE_ALT", "1") == "1":
            # the step decode (opt-in UNSLOTH_AMD_STEP_DECODE=auto | 1): every NF4 weight decoded ONCE per step, in the forward,
            # and kept until its layer's backward (one decoded copy of the projections at the turning point, nothing kept
            # across steps) -- half the decode launches for +12 GB of peak VRAM
            from unsloth_amd import nf4 as _nf4
            _nf4.STEP_DECODE_MODE = "1"
            base_._uamd_auto_policy = None
            alt_point("gc_unsloth_with_step_decode (opt-in UNSLOTH_AMD_STEP_DECODE=auto | 1: NF4 weights decoded once per step, "
                      "kept from a layer's forward to its backward)", "unsloth", B)
            _nf4.STEP_DECODE_MODE = "0"
            base_._uamd_auto_policy = None
            base_._uamd_step_decode = False
            torch.cuda.empty_cache()
        if os.environ.get("BENCH_DP_FORCE_ALT", "1") == "1" and world == 1 and arena is None:
            # the data-parallel path on ONE rank: gradient arena owned by dp.LoRAGradArena, the post-accumulate hooks, the 11
            # bucketed RCCL all-reduces issued from inside the backward (a 1-rank group: the collective is a device copy, its
            # launch / stream-dependency cost is real) -- what DP adds to the step before any xGMI traffic
            def dp_force():
                os.environ["UNSLOTH_AMD_DP_FORCE"] = "1"
                try:
                    if not dist.is_initialized():
                        init_rccl()
                    farena = opt.arena
                    farena._force = True
                    model.for_training(use_gradient_checkpointing=GC_MODE[a.gc])
                    c0 = farena.collectives
                    farena.timing = True
                    farena.exposed_ms()
                    ddt, dpeak, _, dgs = timed_steps(lambda i: training_step(model, batches[i % 2], opt, farena, n_items),
                                                     a.alt_steps, 3, True)
                    exposed = farena.exposed_ms() / (a.alt_steps + 3)
                    farena.timing = False
                    farena._force = False
                    return point(B * T, ddt, a.alt_steps, dpeak, dgs, batch=B, timing="median step x steps",
                                 buckets=len(farena.buckets), bucket_mb=[round((e - s) * 4 / 2**20, 1) for s, e, _ in farena.buckets],
                                 collectives_issued_per_step=(farena.collectives - c0) / (a.alt_steps + 3), rccl_group_size=1,
                                 dp_exposed_ms=round(exposed, 3))
                finally:
                    os.environ["UNSLOTH_AMD_DP_FORCE"] = "0"
            guarded(alt, "dp_path_forced_on_one_rank (UNSLOTH_AMD_DP_FORCE=1: hooks + bucketed RCCL all-reduce inside backward)", dp_force)
        if os.environ.get("BENCH_DECODE_ALT", "1") == "1":
            # single-stream KV-cache decode of the same model through the GEMV kernels, one hipGraph per token
            from unsloth_amd.models.decode import DecodeEngine
            model.eval()
            eng = DecodeEngine(model, max_seq_len=T, batch=1, use_graph=True)
            new = 48
            eng.prefill(torch.randint(0, V, (1, T - new - 8), generator=gi).to(dev))
            tok = torch.zeros(1, dtype=torch.long, device=dev)
            eng.step(tok)
            eng.step(tok)                         # the second call replays the captured graph
            sync()
            t0 = time.time()
            for _ in range(new):
                eng.step(eng.next_tok)
            sync()
            dtok = (time.time() - t0) / new
            cfgm = cfg
            nparam = a.layers * (cfgm.hidden_size * (cfgm.num_attention_heads + 2 * cfgm.num_key_value_heads) * eng.D
                                 + cfgm.hidden_size * cfgm.num_attention_heads * eng.D
                                 + 3 * cfgm.hidden_size * cfgm.intermediate_size)
            hbm_bytes = nparam * 0.516 + V * cfgm.hidden_size * 2 + a.layers * 2 * T * cfgm.num_key_value_heads * eng.D * 2
            from unsloth_amd.models import decode as _md
            alt["decode_batch_1_context_%d (hipGraph per token)" % T] = {
                "tokens_per_s": round(1.0 / dtok, 1), "ms_per_token": round(dtok * 1e3, 3),
                "launches_per_layer": 5 if _md.FUSED_STEP else 14,
                "hbm_bytes_per_token": int(hbm_bytes), "frac_of_hbm_peak": round(hbm_bytes / dtok / 8.0e12, 3)}
            del eng
            torch.cuda.empty_cache()
            model.train()
        if world == 1 and os.environ.get("BENCH_CONFIGS_ALT", "1") == "1":
            # the primary model's 20 GB of decode scratch / caches are not needed below: the other BASELINE configurations
            guarded(alt, C5CE_TAG, lambda: run_config5("ce", a.alt_steps, 3, True))
            guarded(alt, C5LP_TAG, lambda: run_config5("logprob", a.alt_steps, 3, True))
            guarded(alt, C4_TAG, lambda: run_config4(a.alt_steps, 3, True))
            guarded(alt, C1_TAG + "_bs1", lambda: run_config1(1, a.alt_steps, 3, True))
            guarded(alt, C1_TAG + "_bs16", lambda: run_config1(16, a.alt_steps, 3, True))
        if os.environ.get("BENCH_FULLFT_ALT", "1") == "1" and world == 1:
            guarded(alt, "config3_full_finetune_bf16_1gpu (every parameter trains, fp32-master AdamW)",
                    lambda: run_fullft(a.alt_steps, 2))
    opt_name = type(opt).__name__
    if world == 1 and not a.no_gpu_baseline:
        # stock HuggingFace on the SAME box, the SAME batch: what the reference's "x faster than HF" is measured against
        # (/root/reference README.md:87). Our model, optimizer and caches go first: the baseline gets the GPU alone.
        if hasattr(opt, "close"):
            opt.close()
        if arena is not None:
            arena.close()
        del opt, arena, base_
        model = None
        import gc as _gc
        _gc.collect()
        torch.cuda.empty_cache()
        try:
            gpu_base = hf_gpu_baseline(cfg, dev, B, T, a.rank, steps=max(2, a.alt_steps), warmup=2, seed=rank)
        except Exception as ex:
            gpu_base = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            torch.cuda.empty_cache()
        if gpu_base.get("value") and a.alt_steps > 0:
            # ... and with HF's own activation checkpointing: the memory-lean operating point, against `gc_unsloth_min` / `True` above
            try:
                gpu_base["with_gradient_checkpointing"] = hf_gpu_baseline(cfg, dev, B, T, a.rank, steps=max(2, a.alt_steps), warmup=2,
                                                                          seed=rank, checkpointing=True)
            except Exception as ex:
                gpu_base["with_gradient_checkpointing"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
                torch.cuda.empty_cache()
    rccl_ranks = None
    if dist.is_initialized():
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                       # an actual collective over the group: counts the ranks that took part
        rccl_ranks = int(one.item())

    if rank == 0:
        tokens = B * T * a.steps * world
        gem = {k: v for k, v in gs.items() if not k.startswith("attention_")}
        dom = max(gem.values(), key=lambda r: r["total_ms"]) if gem else None
        dom_name = [k for k, v in gem.items() if v is dom][0] if dom else None
        roofline = None
        ms_step = dt / a.steps * 1e3
        if dom:
            # HBM-side bytes per launch from the committed PMC passes of this same command (profiles/pmc_traffic.json,
            # tools/gpu.sh pmc:primary): counters cannot be collected inside a timed run
            traffic, traffic_src = None, None
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
                    pt = json.load(f)
                stem = dom_name.split("<")[0].replace("_kernel", "")      # gemm_nt256: the plain, persistent (p) and NN instances
                ents = [v for k, v in pt.items() if k.startswith(stem) and isinstance(v, dict) and "traffic_bytes" in v]
                if ents:       # the NT and NN template instances of the kernel: dispatch-weighted mean
                    n = sum(max(1, e.get("dispatches", 1)) for e in ents)
                    traffic = int(sum(e["traffic_bytes"] * max(1, e.get("dispatches", 1)) for e in ents) / n)
                    traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE)"
            except (OSError, ValueError, KeyError):
                pass
            fr = fractions(gs, ms_step)
            roofline = dict(bound="mfma", kernel=dom_name, achieved=round(dom["tflops"], 1), peak=MFMA_PEAK_TFLOPS,
                            unit="TFLOP/s", frac=round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                            traffic_unit="bytes per launch (mean)", traffic_source=traffic_src,
                            algorithmic_bytes_per_launch=round(dom["alg_bytes"]),
                            launches_per_step=dom["launches"] // dom["steps_sampled"], avg_launch_us=round(dom["avg_us"], 1),
                            steps_sampled=dom["steps_sampled"],
                            sampling=f"HIP-event pairs around every GEMM and attention launch of every {max(1, a.roofline_every)}th timed step",
                            share_of_step=round(dom["total_ms"] / dom["steps_sampled"] / ms_step, 3),
                            attention={k: v for k, v in fr.items() if k.startswith("attention")},
                            other={k: dict(tflops=round(v["tflops"], 1), avg_us=round(v["avg_us"], 1),
                                           share_of_step=round(v["total_ms"] / v["steps_sampled"] / ms_step, 3))
                                   for k, v in gem.items() if k != dom_name})
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (the other ranks must not wait on it)
            from oracle.cpu_baseline import time_layer
            cpu = time_layer(n_layers=a.layers, budget_s=a.cpu_budget)
            cpu["value"] = round(cpu["value"], 2)
            if os.environ.get("BENCH_CPU_CONFIG1", "1") == "1":
                from oracle.cpu_baseline import time_config1
                cpu["config1_tinyllama_direct"] = time_config1(budget_s=min(15.0, a.cpu_budget))
        vs_gpu = None
        if gpu_base and gpu_base.get("value"):
            vs_gpu = {"tokens_per_s_ratio": round(tokens / dt / gpu_base["value"], 2),
                      "peak_vram_ratio": round(peak / 2**30 / gpu_base["peak_vram_gb"], 2)}
            hgc = gpu_base.get("with_gradient_checkpointing") or {}
            ours = (alt or {}).get("gc_unsloth_min (keep layer inputs only)") or {}
            if hgc.get("value") and ours.get("value"):       # both sides keeping only what a checkpointed layer keeps
                vs_gpu["checkpointed"] = {"ours": "gc_unsloth_min", "tokens_per_s_ratio": round(ours["value"] / hgc["value"], 2),
                                          "peak_vram_ratio": round(ours["peak_vram_gb"] / hgc["peak_vram_gb"], 2)}
        rec = {
            "metric": "train tokens/sec, Llama-3-8B QLoRA (NF4) r=16 seq2048 bf16", "value": round(tokens / dt, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init Llama-3-8B-shaped weights -> NF4, "
            "uniform random token ids)",
            "config": {"workload": "Llama-3-8B QLoRA NF4 r=16 (q,k,v,o,gate,up,down) seq2048 bf16, fwd+bwd+AdamW",
                       "model": "Llama-3-8B (synthetic weights)", "global_batch": B * world, "seq_len": T,
                       "parallelism": f"dp{world}", "layers": a.layers, "lora_rank": a.rank,
                       "gradient_checkpointing": GC_MODE[a.gc], "gc_schedule_chosen": primary_policy,
                       "nf4_decoded_weight_mirrors": ("on (%s GB of bf16 mirrors: no NF4 decode in the step)" % primary_mirror_gb
                                                      if primary_mirrors else
                                                      "off (every NF4 weight is decoded at every use INSIDE the timed step)"),
                       "trainable_params": n_train,
                       "attention": "csrc/attention.hip (causal GQA flash, fwd+bwd)", "optimizer": opt_name + " fp32 on LoRA params"},
            "peak_vram_gb": round(peak / 2**30, 2), "tokens_per_step_per_gpu": B * T, "rccl_ranks": rccl_ranks,
            "loss_first_last": [round(loss_vals[0], 4), round(loss_vals[-1], 4)], "setup_s": round(setup_s, 1),
            "value_with_resident_mirrors": with_mirrors, "vram_batch1_unsloth_min": batch1_min,
            "gpu_baseline": gpu_base,
            "vs_gpu_baseline": vs_gpu,
            "per_rank_ms_per_step": primary_per_rank, "dp": dp_diag,
            "roofline": roofline, "cpu_baseline": cpu, "alt": alt,
        }
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
