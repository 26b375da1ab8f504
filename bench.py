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

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GC_MODE = {"on": True, "off": False, "unsloth": "unsloth", "unsloth:min": "unsloth:min", "unsloth:all": "unsloth:all",
           "unsloth:auto": "unsloth:auto"}
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md (never the 2:1-sparse figure)
HBM_PEAK_GBPS = 8000.0


def llama3_8b_config(n_layers=32, vocab=128256):
    from transformers import LlamaConfig
    return LlamaConfig(
        hidden_size=4096, intermediate_size=14336, num_hidden_layers=n_layers, num_attention_heads=32,
        num_key_value_heads=8, head_dim=128, vocab_size=vocab, rms_norm_eps=1e-5, max_position_embeddings=8192,
        rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, tie_word_embeddings=False,
        attention_bias=False, mlp_bias=False)


KERNEL_OF = {"uamd_gemm_nt_256": "gemm_nt256_kernel<bf16>", "uamd_gemm_nn_256": "gemm_nt256_kernel<bf16>", "uamd_gemm_nt": "gemm_nt_kernel<bf16,dense>",
             "uamd_gemm_nt_nf4": "gemm_nt_kernel<bf16,NF4>"}


class GemmTimer:
    """HIP-event pairs around every MFMA GEMM launch, recorded on the stream the kernel is launched on (torch's
    current stream == the stream passed through the C ABI). One record per launch: (start, end, flops, kernel)."""

    def __init__(self):
        from unsloth_amd.kernels import utils as U
        self.U = U
        self.orig = U._launch_gemm
        self.records = {}
        self.enabled = False
        self.steps_sampled = 0         # timed steps whose launches carry event pairs (bench --roofline-every)

    def install(self):
        U, orig, recs = self.U, self.orig, self.records

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

        U._launch_gemm = timed

    def reset(self):
        self.records.clear()
        self.steps_sampled = 0

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            if not recs:
                continue
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            fl = sum(r[2] for r in recs)
            out[name] = dict(launches=len(recs), steps_sampled=max(1, self.steps_sampled), total_ms=ms,
                             avg_us=ms * 1e3 / len(recs), flops=fl,
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                             alg_bytes=sum(r[3] for r in recs) / len(recs))
        return out


def main():
    # stdout must carry exactly ONE line, the JSON record. Libraries print to the C-level stdout behind Python's back
    # (RCCL's version banner at the first collective, flushed at exit, i.e. AFTER the record): keep the real stdout
    # aside for the record and send everything else written to fd 1 to stderr.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("BENCH_BATCH", 4)), help="sequences per GPU per step")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--gc", choices=["on", "off", "unsloth", "unsloth:min", "unsloth:all", "unsloth:auto"],
                    default=os.environ.get("BENCH_GC", "off"),
                    help="gradient checkpointing for the primary number. off: activations stay in the 288 GB HBM "
                         "(no recompute); on: torch's reentrant per-layer checkpoint (layer inputs only, one extra "
                         "forward per layer); unsloth[:policy]: selective recompute (models/fast_layer.py)")
    ap.add_argument("--alt-steps", type=int, default=int(os.environ.get("BENCH_ALT_STEPS", 3)),
                    help="also time this many steps in the other checkpointing modes and at batch 1 / 2 "
                         "(reported under 'alt'); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-every", type=int, default=int(os.environ.get("BENCH_ROOFLINE_EVERY", 4)),
                    help="HIP-event pairs around the GEMM launches on every Nth timed step (the first one always). An event "
                         "pair drains the queue around its kernel: ~12 us per GEMM, 292 GEMMs per step = 1.4 %% of the step "
                         "when every step is instrumented (profiles/r03final_step_sequence.csv: all 3.5 ms of idle gaps of a "
                         "step sit before a GEMM or before the kernel that follows one). 1 = every step")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and os.environ.get("BENCH_ALT_MULTI", "0") != "1":
        a.alt_steps = 0            # the other operating points are a 1-GPU report; a scaling run times the primary only
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda unavailable); the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    force_dp = os.environ.get("UNSLOTH_AMD_DP_FORCE", "0") == "1"     # 1-rank RCCL group: exercises the DP path on one GPU
    if world > 1 or force_dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one node over xGMI: RCCL's bootstrap needs no NIC; keep it off interface / InfiniBand probing
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        if force_dp and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)          # "nccl" IS RCCL on ROCm
    if a.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)

    from unsloth_amd import FastLanguageModel
    from unsloth_amd.dp import LoRAGradArena
    from unsloth_amd.trainer import make_optimizer, training_step

    cfg = llama3_8b_config(a.layers)
    t_setup = time.time()
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=a.seq, dtype=torch.bfloat16,
                                                 load_in_4bit=True, device=dev, random_state=3407,
                                                 use_gradient_checkpointing=GC_MODE[a.gc])
    # NOTE: for_training() below re-applies the checkpointing mode per measurement
    model = FastLanguageModel.get_peft_model(model, r=a.rank, lora_alpha=a.rank, lora_dropout=0.0, bias="none",
                                             use_gradient_checkpointing=GC_MODE[a.gc], random_state=3407)
    g = torch.Generator(device="cpu").manual_seed(3407)
    n_train = 0
    for n, p in model.named_parameters():
        if p.requires_grad:
            n_train += p.numel()
            if "lora_B" in n:           # non-zero B so every LoRA gradient is exercised with real numbers
                p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
    arena = LoRAGradArena(model) if (world > 1 or force_dp) else None
    # optim.FlatAdamW: parameters / gradients / moments in flat arenas, one launch per step (the gradients live in the
    # data-parallel arena when there is one; UNSLOTH_AMD_FLAT_ADAMW=0 = torch's fused AdamW)
    opt = make_optimizer(model, lr=2e-4, arena=arena)
    B, T, V = a.batch, a.seq, cfg.vocab_size
    gi = torch.Generator(device="cpu").manual_seed(rank)       # different data per rank

    def make_batches(bs):
        out = []
        for _ in range(2):
            ids = torch.randint(0, V, (bs, T), generator=gi).to(dev)
            pos = torch.arange(T, dtype=torch.int32, device=dev).unsqueeze(0).expand(bs, T).contiguous()
            out.append(dict(input_ids=ids, labels=ids.clone(), position_ids=pos))
        return out, torch.tensor((T - 1) * bs * world, device=dev)      # global non-ignored targets per step
    batches, n_items = make_batches(B)
    timer = GemmTimer()
    timer.install()
    setup_s = time.time() - t_setup

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(gc_mode, steps, warmup, data=None, per_step=False):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks.
        per_step (the short `alt` points only, never the primary): every step is bracketed on its own and the MEDIAN step
        time x K is returned -- a 3-step point is otherwise at the mercy of one allocator stall after empty_cache()."""
        bt, ni = data if data is not None else (batches, n_items)
        model.for_training(use_gradient_checkpointing=gc_mode)
        losses = []
        for i in range(warmup):
            losses.append(training_step(model, bt[i % 2], opt, arena, ni))
        sync()
        torch.cuda.reset_peak_memory_stats()
        timer.reset()
        timer.enabled = True
        t0 = time.perf_counter()
        if per_step:
            times = []
            for i in range(steps):
                ts = time.perf_counter()
                timer.enabled = (i == 0)                # the MEDIAN step is then one without event pairs
                timer.steps_sampled += int(timer.enabled)
                losses.append(training_step(model, bt[i % 2], opt, arena, ni))
                sync()
                times.append(time.perf_counter() - ts)
            dt = sorted(times)[len(times) // 2] * steps
        else:
            every = max(1, a.roofline_every)
            for i in range(steps):
                timer.enabled = (i % every == 0)        # roofline sample: the launches of every Nth timed step
                timer.steps_sampled += int(timer.enabled)
                losses.append(training_step(model, bt[i % 2], opt, arena, ni))
            sync()
            dt = time.perf_counter() - t0
        timer.enabled = False
        peak = torch.cuda.max_memory_allocated()
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
            pk = torch.tensor([peak], device=dev, dtype=torch.int64)
            dist.all_reduce(pk, op=dist.ReduceOp.MAX)
            peak = int(pk)
        return dt, peak, [float(l) for l in losses], timer.summary()

    # the PRIMARY measurement first, on the freshly initialised model (loss_first_last starts at ~ln V); the other operating
    # points afterwards (their warm-up / timing steps keep training the same adapters, which no longer matters)
    dt, peak, loss_vals, gs = measure(GC_MODE[a.gc], a.steps, a.warmup)
    alt = None
    if a.alt_steps > 0:
        # the other checkpointing modes at the primary batch, then batch 1 and 2 without checkpointing
        alt = {}

        def alt_point(tag, gc_mode, bs):
            data = None if bs == B else make_batches(bs)
            adt, apeak, _, ags = measure(gc_mode, a.alt_steps, 3, data, per_step=True)     # 3 warm-up steps: allocator growth after empty_cache(), decoded mirrors
            dom_ = max(ags.values(), key=lambda r: r["total_ms"]) if ags else None
            alt[tag] = {"gradient_checkpointing": gc_mode, "batch": bs, "steps": a.alt_steps, "timing": "median step x steps",
                        "value": round(bs * T * a.alt_steps * world / adt, 1),
                        "ms_per_step": round(adt / a.alt_steps * 1e3, 2), "peak_vram_gb": round(apeak / 2**30, 2),
                        "gemm_tflops": round(dom_["tflops"], 1) if dom_ else None,
                        "gemm_frac_of_mfma_peak": round(dom_["tflops"] / MFMA_PEAK_TFLOPS, 4) if dom_ else None}
            del data
            torch.cuda.empty_cache()
        for tag, mode in (("gc_torch_reentrant (reference's True)", True),
                          ("gc_unsloth_selective_recompute (keep attention block, re-run gate/up)", "unsloth"),
                          ("gc_unsloth_auto (as many keep-everything layers as the free HBM holds)", "unsloth:auto"),
                          ("gc_unsloth_min (keep layer inputs only)", "unsloth:min"), ("gc_off", False)):
            if mode != GC_MODE[a.gc]:
                alt_point(tag, mode, B)
        for bs in (1, 2):
            if bs != B:
                alt_point(f"batch_{bs}_gc_off", False, bs)
        if os.environ.get("BENCH_RESIDENT_ALT", "1") == "1":
            # opt-in mode: decoded bf16 mirrors of the NF4 weights stay in HBM (+2 B/param), no decode launches
            from unsloth_amd import nf4 as _nf4
            _nf4.set_resident(True)
            alt_point("weights_resident_bf16_gc_off (opt-in: UNSLOTH_AMD_RESIDENT_WEIGHTS=1)", False, B)
            alt_point("weights_resident_bf16_batch_1", False, 1)
            _nf4.set_resident(False)
            torch.cuda.empty_cache()
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
            alt["decode_batch_1_context_%d (hipGraph per token)" % T] = {
                "tokens_per_s": round(1.0 / dtok, 1), "ms_per_token": round(dtok * 1e3, 3),
                "hbm_bytes_per_token": int(hbm_bytes), "frac_of_hbm_peak": round(hbm_bytes / dtok / 8.0e12, 3)}
            del eng
            torch.cuda.empty_cache()
            model.train()
        if os.environ.get("BENCH_FULLFT_ALT", "1") == "1" and world == 1:
            # BASELINE config 3 on ONE GPU: the same architecture fully trainable in bf16 (dense dW GEMMs, norm / lm_head /
            # embedding gradients, flat buckets, fp32-master AdamW: 16 + 16 + 96 GB of the 288), world size 1 = no collective
            try:
                from unsloth_amd.full_finetune import ShardedAdamW, full_finetune_step
                timer.enabled = False
                fmodel, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=a.seq, dtype=torch.bfloat16,
                                                              full_finetuning=True, device=dev, random_state=3407,
                                                              use_gradient_checkpointing=False)
                fopt = ShardedAdamW(fmodel, lr=1e-5)
                fl = []
                for i in range(2):
                    fl.append(full_finetune_step(fmodel, batches[i % 2], fopt, n_items))
                sync()
                torch.cuda.reset_peak_memory_stats()
                ft = []
                for i in range(a.alt_steps):
                    ts = time.perf_counter()
                    fl.append(full_finetune_step(fmodel, batches[i % 2], fopt, n_items))
                    sync()
                    ft.append(time.perf_counter() - ts)
                fdt = sorted(ft)[len(ft) // 2]
                n_all = sum(p.numel() for p in fmodel.parameters())
                alt["config3_full_finetune_bf16_1gpu (every parameter trains, fp32-master AdamW)"] = {
                    "batch": B, "steps": a.alt_steps, "timing": "median step", "value": round(B * T / fdt, 1),
                    "ms_per_step": round(fdt * 1e3, 2), "peak_vram_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                    "trainable_params": n_all, "model_tflops_per_s": round(6.0 * n_all * B * T / fdt / 1e12, 1),
                    "loss_first_last": [round(float(fl[0]), 4), round(float(fl[-1]), 4)]}
                fopt.buckets.close()
                del fmodel, fopt
                torch.cuda.empty_cache()
            except Exception as ex:                     # an operating point must never take the primary record down with it
                alt["config3_full_finetune_bf16_1gpu"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
                torch.cuda.empty_cache()
            os.environ["UNSLOTH_ENABLE_FULL_FINETUNING"] = "0"
    rccl_ranks = None
    if dist.is_initialized():
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                       # an actual collective over the group: counts the ranks that took part
        rccl_ranks = int(one.item())

    if rank == 0:
        tokens = B * T * a.steps * world
        dom = max(gs.values(), key=lambda r: r["total_ms"]) if gs else None
        dom_name = [k for k, v in gs.items() if v is dom][0] if dom else None
        roofline = None
        if dom:
            # HBM-side bytes per launch from the committed PMC passes of this same command (profiles/pmc_traffic.json,
            # tools/gpu_pmc_bench.sh): counters cannot be collected inside a timed run
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
            roofline = dict(bound="mfma", kernel=dom_name, achieved=round(dom["tflops"], 1), peak=MFMA_PEAK_TFLOPS,
                            unit="TFLOP/s", frac=round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                            traffic_unit="bytes per launch (mean)", traffic_source=traffic_src,
                            algorithmic_bytes_per_launch=round(dom["alg_bytes"]),
                            launches_per_step=dom["launches"] // dom["steps_sampled"], avg_launch_us=round(dom["avg_us"], 1),
                            steps_sampled=dom["steps_sampled"],
                            sampling=f"HIP-event pairs around every GEMM launch of every {max(1, a.roofline_every)}th timed step",
                            share_of_step=round(dom["total_ms"] / dom["steps_sampled"] / (dt / a.steps * 1e3), 3),
                            other={k: dict(tflops=round(v["tflops"], 1), avg_us=round(v["avg_us"], 1),
                                           share_of_step=round(v["total_ms"] / v["steps_sampled"] / (dt / a.steps * 1e3), 3))
                                   for k, v in gs.items() if k != dom_name})
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (the other ranks must not wait on it)
            from oracle.cpu_baseline import time_layer
            cpu = time_layer(n_layers=a.layers, budget_s=a.cpu_budget)
            cpu["value"] = round(cpu["value"], 2)
            if os.environ.get("BENCH_CPU_CONFIG1", "1") == "1":
                from oracle.cpu_baseline import time_config1
                cpu["config1_tinyllama_direct"] = time_config1(budget_s=min(15.0, a.cpu_budget))
        rec = {
            "metric": "train tokens/sec, Llama-3-8B QLoRA (NF4) r=16 seq2048 bf16", "value": round(tokens / dt, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init Llama-3-8B-shaped weights -> NF4, "
            "uniform random token ids)",
            "config": {"workload": "Llama-3-8B QLoRA NF4 r=16 (q,k,v,o,gate,up,down) seq2048 bf16, fwd+bwd+AdamW",
                       "model": "Llama-3-8B (synthetic weights)", "global_batch": B * world, "seq_len": T,
                       "parallelism": f"dp{world}", "layers": a.layers, "lora_rank": a.rank,
                       "gradient_checkpointing": GC_MODE[a.gc], "trainable_params": n_train,
                       "attention": "csrc/attention.hip (causal GQA flash, fwd+bwd)", "optimizer": type(opt).__name__ + " fp32 on LoRA params"},
            "peak_vram_gb": round(peak / 2**30, 2), "tokens_per_step_per_gpu": B * T, "rccl_ranks": rccl_ranks,
            "loss_first_last": [round(loss_vals[0], 4), round(loss_vals[-1], 4)], "setup_s": round(setup_s, 1),
            "roofline": roofline, "cpu_baseline": cpu, "alt": alt,
        }
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
