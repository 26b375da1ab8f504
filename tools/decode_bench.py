"""Single-stream decode throughput of the Llama-3-8B-shaped QLoRA model (NF4 r=16) with the hipGraph-replayed step
(models/decode.py), and the GEMV kernels alone against the HBM roofline.

    python tools/decode_bench.py [--layers 32] [--context 2048] [--new 64] [--out gpurun_out/decode.jsonl]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM = 8.0e12


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--context", type=int, default=2048)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="whole model, hipGraph step only (for a rocprofv3 kernel trace)")
    a = ap.parse_args()
    out = open(a.out, "w") if a.out else None

    def emit(d):
        print(json.dumps(d), flush=True)
        if out:
            out.write(json.dumps(d) + "\n")

    from unsloth_amd.kernels import decode as D
    from unsloth_amd.nf4 import quantize_nf4
    dev, bf = "cuda", torch.bfloat16
    # ---- GEMV kernels alone (Llama-3-8B shapes)
    for name, Ns, K in () if a.quick else (("q|k|v", (4096, 1024, 1024), 4096), ("o", (4096,), 4096), ("gate|up", (14336, 14336), 4096),
                        ("down", (4096,), 14336)):
        x = torch.randn(K, device=dev, dtype=bf)
        projs = []
        for N in Ns:
            W = (torch.randn(N, K, device=dev) * 0.02).to(bf)
            packed, qs = quantize_nf4(W, compress_statistics=True)
            qs.dtype = bf
            projs.append((packed, qs, None, None, None))
        outbuf = torch.empty(sum(Ns), device=dev, dtype=bf)
        us = timeit(lambda: D.linear_group(x, projs, out=outbuf))
        byts = sum(Ns) * K * (0.5 + 1 / 64 + 4 / (64 * 256)) + K * 2 + sum(Ns) * 2
        emit(dict(kernel=f"gemv_nf4 {name}", us=round(us, 2), bytes=int(byts), GBps=round(byts / us / 1e3, 1),
                  frac_hbm=round(byts / (us * 1e-6) / HBM, 3)))
    # ---- the five launches of one fused decoder-layer step, each alone (LoRA r=16 on every projection, t = A x inside)
    from unsloth_amd.kernels.decode import attn_decode_fused, attn_decode, rope_kv_append
    H, I, Hq, Hk, Dh = 4096, 14336, 32, 8, 128

    def lora_projs(Ns, K):
        ps = []
        for N in Ns:
            W = (torch.randn(N, K, device=dev) * 0.02).to(bf)
            packed, qs = quantize_nf4(W, compress_statistics=True)
            qs.dtype = bf
            ps.append((packed, qs, torch.nn.Parameter(torch.randn(16, K, device=dev) * 0.02),
                       torch.nn.Parameter(torch.randn(N, 16, device=dev) * 0.02), 1.0, None))
        return ps
    resid = torch.randn(H, device=dev, dtype=bf)
    delta = torch.randn(H, device=dev, dtype=bf)
    wn = torch.ones(H, device=dev, dtype=bf)
    hbuf = torch.empty(H, device=dev, dtype=bf)
    for name, Ns, K, fused, xin in () if a.quick else (
            ("q|k|v  (add + RMSNorm in)", (4096, 1024, 1024), H, dict(mode=2, res=resid, norm_w=wn, eps=1e-5, h_out=hbuf), delta),
            ("o", (4096,), H, dict(mode=0), delta),
            ("gate|up (add + RMSNorm in, SwiGLU out)", (I, I), H, dict(mode=2, res=resid, norm_w=wn, eps=1e-5, h_out=hbuf, glu=True), delta),
            ("down", (4096,), I, dict(mode=0), torch.randn(I, device=dev, dtype=bf))):
        projs = lora_projs(Ns, K)
        outbuf = torch.empty(Ns[0] if fused.get("glu") else sum(Ns), device=dev, dtype=bf)
        us = timeit(lambda: D.linear_group(xin, projs, out=outbuf, fused=fused))
        us14 = timeit(lambda: D.linear_group(xin, projs))
        byts = sum(Ns) * K * (0.5 + 1 / 64 + 4 / (64 * 256)) + K * 2 + sum(Ns) * 2
        emit(dict(kernel=f"fused gemv_nf4 {name}", us=round(us, 2), separate_t_plus_gemv_us=round(us14, 2), bytes=int(byts),
                  GBps=round(byts / us / 1e3, 1), frac_hbm=round(byts / (us * 1e-6) / HBM, 3)))
    S = a.context if not a.quick else 128
    kc = torch.randn(1, Hk, S, Dh, device=dev, dtype=bf)
    vc = torch.randn(1, Hk, S, Dh, device=dev, dtype=bf)
    qkv = torch.randn(1, (Hq + 2 * Hk) * Dh, device=dev, dtype=bf)
    cos = torch.randn(S, Dh // 2, device=dev, dtype=bf)
    sin = torch.randn(S, Dh // 2, device=dev, dtype=bf)
    kvl = torch.full((1,), S - 1, dtype=torch.int32, device=dev)
    part = torch.empty(1, Hq, S // 128, Dh + 2, dtype=torch.float32, device=dev)
    fpart, cnt = D.fused_attn_workspace(1, Hq, Hk, S, Dh, 128, dev)
    ao = torch.empty(1, Hq * Dh, device=dev, dtype=bf)
    us = timeit(lambda: attn_decode_fused(qkv, cos, sin, kvl, kc, vc, ao, fpart, cnt, 128, 0.088, Hq))

    def three():
        rope_kv_append(qkv, cos, sin, kvl, kc, vc, Hq, Hk, Dh)
        attn_decode(qkv[:, :Hq * Dh], kc, vc, kvl, ao, part, 128, 0.088)
    us3 = timeit(three)
    byts = 2 * S * Hk * Dh * 2
    emit(dict(kernel=f"fused attention (rope + append + split-KV + combine), context {S}", us=round(us, 2),
              three_launches_us=round(us3, 2), bytes=byts, GBps=round(byts / us / 1e3, 1), frac_hbm=round(byts / (us * 1e-6) / HBM, 3)))
    del kc, vc
    V, K = 128256, 4096
    W = (torch.randn(V, K, device=dev) * 0.02).to(bf)
    x = torch.randn(K, device=dev, dtype=bf)
    us = timeit(lambda: D.gemv(x, [dict(W=W, N=V, y_f32=True)], nf4=False))
    byts = V * K * 2 + V * 4
    if not a.quick:
        emit(dict(kernel="gemv_bf16 lm_head", us=round(us, 2), bytes=byts, GBps=round(byts / us / 1e3, 1),
              frac_hbm=round(byts / (us * 1e-6) / HBM, 3)))
    del W

    # ---- whole model
    import bench as B                                   # the benchmark's synthetic Llama-3-8B-shaped config
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.models import decode as MD
    from unsloth_amd.models.decode import DecodeEngine
    cfg = B.llama3_8b_config(a.layers)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=a.context, dtype=torch.bfloat16, load_in_4bit=True,
                                                 device=torch.device(dev), random_state=3407, use_gradient_checkpointing=False)
    model = FastLanguageModel.get_peft_model(model, r=16, lora_alpha=16, lora_dropout=0.0, bias="none",
                                             use_gradient_checkpointing=False, random_state=3407)
    gg = torch.Generator(device="cpu").manual_seed(3407)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=gg) * 0.02).to(p.device))
    model.eval()
    ids = torch.randint(0, 32000, (1, a.context - a.new), device=dev)
    for fused, graph in ((True, True),) if a.quick else ((True, True), (False, True), (True, False), (False, False)):
        MD.FUSED_STEP = fused
        eng = DecodeEngine(model, max_seq_len=a.context, batch=1, use_graph=graph)
        t0 = time.time()
        eng.prefill(ids)
        torch.cuda.synchronize()
        t_prefill = time.time() - t0
        tok = torch.zeros(1, dtype=torch.long, device=dev)
        eng.step(tok)
        eng.step(tok)                                   # warm: the second call replays the graph
        torch.cuda.synchronize()
        n = a.new - 4
        t0 = time.time()
        for _ in range(n):
            eng.step(eng.next_tok)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        # HBM bytes one token must read: NF4 projections + lm_head + KV cache at this context
        cfg = eng.cfg
        nparam = a.layers * (cfg.hidden_size * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * eng.D
                             + cfg.hidden_size * cfg.num_attention_heads * eng.D + 3 * cfg.hidden_size * cfg.intermediate_size)
        byts = nparam * 0.516 + cfg.vocab_size * cfg.hidden_size * 2 + a.layers * 2 * a.context * cfg.num_key_value_heads * eng.D * 2
        emit(dict(metric="decode tokens/s, batch 1", launches_per_layer=5 if fused else 14, hipgraph=graph, layers=a.layers, context=a.context, ms_per_token=round(dt * 1e3, 3),
                  tokens_per_s=round(1 / dt, 1), prefill_s=round(t_prefill, 3), hbm_bytes_per_token=int(byts),
                  frac_hbm=round(byts / dt / HBM, 3)))
        del eng


if __name__ == "__main__":
    main()
