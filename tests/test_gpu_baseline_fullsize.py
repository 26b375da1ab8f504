"""-m gpu: the BASELINE.json configurations as parity cases AT THEIR STATED SIZES: real widths, the
stated sequence length, the stated quantisation / rank, >= 2 decoder layers, every gradient-checkpointing mode -- the whole
drop-in surface (from_pretrained -> get_peft_model -> forward / backward) on the HIP path against the
implementation-independent oracle: stock HuggingFace modules in fp32 over oracle-dequantised NF4 weights + merged LoRA
(oracle/ref_model.py; run on the GPU in fp32 here so that 8B-wide layers at 2048-4096 tokens take seconds -- torch's own
fp32 GEMMs and eager attention, no kernel of the product).

  config 2  Llama-3-8B widths, QLoRA NF4 r=16, seq 2048, batch 1 and 4, use_gradient_checkpointing False / "unsloth" / True
  config 4  Qwen2-VL-7B language tower (3584 / 18944 / 28:4 heads / vocab 152064, q/k/v BIAS), NF4 + LoRA r=32, seq 4096,
            [3, B, T] multimodal positions, against transformers' Qwen2VLTextModel; modes False and "unsloth"
  config 5  Mistral-7B widths, NF4 + LoRA r=16, seq 4096 (sliding window 4096: inactive, mistral.py:116-120), the causal-LM
            loss AND the chunked per-token log-prob leg of the GRPO / DPO runs on lm_head [32000, 4096]
  (config 1 -- TinyLlama, seq 512 -- is already at its stated size in tests/test_gpu_baseline_configs.py; config 3 -- full
   fine-tuning -- in tests/test_gpu_full_finetune.py)

Bounds: loss within 1e-3 (north star). LoRA gradients, relative Frobenius against the fp32 oracle: every tensor within
2.5e-2 and all of them together within 1.5e-2 (2x the 1.2e-2 measured at small sizes) -- or, where the bf16 rounding noise
of 8B-wide layers over 2048-4096 tokens exceeds that by itself, within 1.25x the error STOCK HuggingFace in bf16 has against
the same oracle on the same inputs (the yardstick run, oracle/ref_model.py `dtype=`), and never beyond 5e-2 / 3e-2.
Measured values: profiles/r03_fullsize_parity.json.
"""
import json
import os

import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
REPORT = {}
LOSS_TOL, WORST_TOL, TOTAL_TOL = 1e-3, 2.5e-2, 1.5e-2


def _report(name, **kw):
    REPORT[name] = kw
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fullsize_parity.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _build(cfg, r, max_seq, fast_model=False):
    from unsloth_amd import FastLanguageModel, FastModel
    cls = FastModel if fast_model else FastLanguageModel
    model, _ = cls.from_pretrained(config=cfg, max_seq_length=max_seq, load_in_4bit=True, device=DEV, random_state=3407,
                                   use_gradient_checkpointing=False)
    model = FastLanguageModel.get_peft_model(model, r=r, lora_alpha=r, use_gradient_checkpointing=False, random_state=3407)
    g = torch.Generator().manual_seed(3407)
    for n, p in model.named_parameters():
        if "lora_B" in n:                      # PEFT's default B = 0 would zero half of the gradients
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(DEV))
    return model


def _grads(model):
    return {"layers." + n.split(".layers.", 1)[1].replace(".default.weight", ""): p.grad.detach().float().cpu()
            for n, p in model.named_parameters() if p.requires_grad}


def _zero(model):
    for p in model.parameters():
        p.grad = None


def _errors(got, ref):
    worst = max(rel_fro(got[k], ref[k]) for k in got)
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]), torch.cat([ref[k].flatten() for k in sorted(got)]))
    return worst, total


def _compare(name, loss, got, ref_loss, ref, yard=None):
    """Loss within 1e-3 of the fp32 oracle. LoRA gradients: within the absolute bound (WORST_TOL / TOTAL_TOL) -- or, where
    the bf16 noise of 8B-wide layers over thousands of tokens exceeds it, never more than 1.25x further from the fp32 truth
    than STOCK HuggingFace run in bf16 on the same inputs (`yard`: its gradients), and never beyond 2x the absolute bound."""
    dl = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    assert set(got) == set(ref)
    worst, total = _errors(got, ref)
    rec = dict(loss=float(loss), oracle_loss=float(ref_loss), loss_rel_err=dl, worst_grad_rel_fro=worst,
               total_grad_rel_fro=total)
    worst_tol, total_tol = WORST_TOL, TOTAL_TOL
    if yard is not None:
        yw, yt = _errors(yard, ref)
        rec.update(hf_bf16_worst_grad_rel_fro=yw, hf_bf16_total_grad_rel_fro=yt)
        worst_tol = min(2 * WORST_TOL, max(WORST_TOL, 1.25 * yw))
        total_tol = min(2 * TOTAL_TOL, max(TOTAL_TOL, 1.25 * yt))
    _report(name, **rec)
    assert dl <= LOSS_TOL, (name, float(loss), float(ref_loss))
    assert worst < worst_tol and total < total_tol, (name, worst, total, worst_tol, total_tol)


# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def llama3_8b_two_layers():
    from transformers import LlamaConfig
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=8, head_dim=128, vocab_size=128256, rms_norm_eps=1e-5, max_position_embeddings=8192,
                      rope_parameters={"rope_type": "llama3", "rope_theta": 5e5, "factor": 8.0, "low_freq_factor": 1.0,
                                       "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
                      tie_word_embeddings=False)
    model = _build(cfg, r=16, max_seq=2048)
    yield model
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("batch", [1, 4])
def test_config2_llama3_8b_qlora_r16_seq2048_every_checkpointing_mode(llama3_8b_two_layers, batch):
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    from unsloth_amd import FastLanguageModel
    model = llama3_8b_two_layers
    assert model.get_base_model()._unsloth_amd_patched == (2, 2, 2), "fused hooks not installed on every layer"
    T = 2048
    g = torch.Generator().manual_seed(batch)
    ids = torch.randint(0, 128256, (batch, T), generator=g)
    labels = ids.clone()
    labels[0, :11] = -100
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0).expand(batch, T).contiguous()
    ref_loss, ref = hf_reference_loss_and_lora_grads(model, ids, labels, pos, device="cuda")
    torch.cuda.empty_cache()
    _, yard = hf_reference_loss_and_lora_grads(model, ids, labels, pos, device="cuda", dtype=torch.bfloat16)
    torch.cuda.empty_cache()
    seen = {}
    for mode in (False, "unsloth", True):
        FastLanguageModel.for_training(model, use_gradient_checkpointing=mode)
        _zero(model)
        out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
        out.loss.backward()
        got = _grads(model)
        _compare(f"config2_b{batch}_gc_{mode}", out.loss, got, ref_loss, ref, yard)
        seen[mode] = (float(out.loss), got)
    # the three modes run the same kernels on the same inputs: selective recompute is BITWISE the no-checkpoint result
    assert seen[False][0] == seen["unsloth"][0]
    assert all(torch.equal(seen[False][1][k], seen["unsloth"][1][k]) for k in seen[False][1])
    _zero(model)


def test_default_unsloth_spelling_fits_the_free_hbm(llama3_8b_two_layers):
    """`use_gradient_checkpointing="unsloth"` (the API default) = the least-recompute schedule that fits (ref: the mode is a
    fit-to-memory decision, models/_utils.py:360-386). On the idle GPU it keeps everything: the peak of `False`. With most of
    the HBM held by somebody else (a reserved dummy tensor, no environment switch) the same spelling falls back to the
    keep-attention policy: the peak of "unsloth:attn". auto_schedule's transient estimate must be conservative -- the largest
    free size at which it still answers "attn" has to cover what the "attn" step really takes, or the fallback would OOM."""
    from unsloth_amd import FastLanguageModel, nf4
    from unsloth_amd.models import fast_layer as F
    model = llama3_8b_two_layers
    # UNSLOTH_AMD_RESIDENT_WEIGHTS=auto for this test (the default is "0" since round 5): the mirrors ride on the same decision
    mode_was, nf4.RESIDENT_MODE = nf4.RESIDENT_MODE, "auto"
    try:
        _default_spelling_body(model, FastLanguageModel, F)
    finally:
        nf4.RESIDENT_MODE = mode_was
        nf4.set_resident(False, model=model.get_base_model().model)


def _default_spelling_body(model, FastLanguageModel, F):
    T = 2048
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 128256, (4, T), generator=g).to(DEV)
    pos = torch.arange(T, dtype=torch.int32, device=DEV).unsqueeze(0).expand(4, T).contiguous()
    seen = {}

    def step(mode):
        FastLanguageModel.for_training(model, use_gradient_checkpointing=mode)
        model.get_base_model().model._uamd_auto_policy = None
        _zero(model)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = model(input_ids=ids, labels=ids, position_ids=pos)
        out.loss.backward()
        torch.cuda.synchronize()
        pol = model.get_base_model().model._uamd_auto_policy
        return torch.cuda.max_memory_allocated() - base, float(out.loss), _grads(model), (pol[1] if pol else None)

    step("unsloth:attn")                                       # warm-up: per-device scratch is allocated on first use
    for mode in (False, "unsloth:attn", "unsloth"):
        seen[mode] = step(mode)
    assert seen["unsloth"][3] == F.POLICIES["all"], "an idle 288 GB part must keep everything"
    # ... and (nf4.RESIDENT_MODE "auto") the decoded 16-bit mirrors of the NF4 projections with it: 2 B per parameter, allocated
    # during that first step
    from unsloth_amd import nf4
    mirrors = 2 * (4096 * 6144 + 4096 * 4096 + 3 * 4096 * 14336) * 2
    inner = model.get_base_model().model
    assert nf4.resident_count(inner) == 14 and inner._uamd_mirrors_auto and not nf4.RESIDENT, nf4.resident_count(inner)
    assert seen["unsloth"][0] <= 1.05 * seen[False][0] + mirrors and seen["unsloth:attn"][0] < 0.8 * seen[False][0], seen
    # the largest free size at which the arithmetic still says "attn everywhere" for this model and batch
    kw = dict(n_layers=2, tokens=4 * T, hidden=4096, inter=14336, qkv_cols=6144, elsize=2, vocab=128256)
    lo, hi = 0, 64 << 30
    while hi - lo > (1 << 20):
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if F.auto_schedule(free_bytes=mid, **kw) == F.POLICIES["attn"] else (lo, mid)
    assert lo > seen["unsloth:attn"][0] + (64 << 20), ("auto_schedule would plan an 'attn' step into less memory than it takes",
                                                        lo, seen["unsloth:attn"][0])
    _zero(model)
    torch.cuda.empty_cache()
    free = F.free_hbm_bytes(DEV)
    dummy = torch.empty(free - lo + (8 << 20), dtype=torch.uint8, device=DEV)       # somebody else's memory
    try:
        assert F.free_hbm_bytes(DEV) <= lo
        crowded = step("unsloth")
    finally:
        del dummy
        torch.cuda.empty_cache()
    assert crowded[3] == F.POLICIES["attn"]
    assert nf4.resident_count(inner) == 0 and not inner._uamd_mirrors_auto     # the mirrors went first
    assert crowded[0] <= 1.02 * seen["unsloth:attn"][0], (crowded[0], seen["unsloth:attn"][0])
    # and it is the same step: every gradient bitwise (the loss itself is summed over fused-CE row chunks whose size follows
    # the free memory too -- another summation order, the last bit of the fp32 sum may differ)
    assert abs(crowded[1] - seen[False][1]) <= 1e-6 * abs(seen[False][1])
    assert all(torch.equal(crowded[2][k], seen[False][2][k]) for k in crowded[2])
    _report("default_unsloth_spelling", idle_peak_gb=seen["unsloth"][0] / 2**30, no_gc_peak_gb=seen[False][0] / 2**30,
            attn_peak_gb=seen["unsloth:attn"][0] / 2**30, crowded_peak_gb=crowded[0] / 2**30, largest_attn_free_gb=lo / 2**30)
    _zero(model)


# ------------------------------------------------------------------------------------------------------------------
def test_config5_mistral_7b_lora_r16_seq4096_loss_and_logprob_leg():
    from transformers import MistralConfig
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    from unsloth_amd.models.rl_replacements import chunked_hidden_states_selective_log_softmax
    cfg = MistralConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=2, num_attention_heads=32,
                        num_key_value_heads=8, head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, max_position_embeddings=32768,
                        sliding_window=4096, rope_parameters={"rope_type": "default", "rope_theta": 1e4},
                        tie_word_embeddings=False)
    model = _build(cfg, r=16, max_seq=4096)
    assert model.get_base_model()._unsloth_amd_patched == (2, 2, 2)
    T, V = 4096, 32000
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (1, T), generator=g)
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0)
    # --- leg 1: causal-LM loss (the fused linear-CE path), window inactive at seq 4096
    out = model(input_ids=ids.to(DEV), labels=ids.to(DEV), position_ids=pos.to(DEV))
    out.loss.backward()
    ref_loss, ref = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), pos, device="cuda")
    _, yard = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), pos, device="cuda", dtype=torch.bfloat16)
    _compare("config5_ce", out.loss, _grads(model), ref_loss, ref, yard)
    _zero(model)
    torch.cuda.empty_cache()
    # --- leg 2: per-token log-probs of the next token, chunked over lm_head [32000, 4096] (GRPO / DPO), with a
    #     completion mask and per-token weights standing in for advantages; forward values AND LoRA gradients
    mask = torch.zeros(1, T - 1)
    mask[0, T // 2:] = 1.0                                   # "completion" = second half
    wts = torch.randn(1, T - 1, generator=g) * mask
    nxt = ids[:, 1:]
    box = {}

    def objective(logits):                                    # fp32 oracle: log_softmax(logits)[next token]
        lp = torch.log_softmax(logits[:, :-1].float(), dim=-1).gather(-1, nxt.to(logits.device).unsqueeze(-1)).squeeze(-1)
        box["lp"] = lp.detach().cpu()
        return -(lp * wts.to(lp.device)).sum() / mask.sum()

    ref_obj, ref2 = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), pos, device="cuda", loss_fn=objective)
    os.environ["UNSLOTH_RETURN_HIDDEN_STATES"] = "1"
    try:
        hidden = model(input_ids=ids.to(DEV), position_ids=pos.to(DEV)).logits        # hidden states in the logits slot
    finally:
        os.environ["UNSLOTH_RETURN_HIDDEN_STATES"] = "0"
    assert hidden.shape == (1, T, 4096)
    lm_head = model.get_base_model().lm_head.weight
    lp = chunked_hidden_states_selective_log_softmax(hidden[:, :-1], lm_head, nxt.to(DEV), chunks=4)
    oracle_lp = box["lp"]
    err = (lp.detach().cpu() - oracle_lp).abs().max().item()
    rel = ((lp.detach().cpu() - oracle_lp).norm() / oracle_lp.norm()).item()
    scale = oracle_lp.abs().max().item() + 1.0
    obj = -(lp * wts.to(DEV)).sum() / mask.sum().to(DEV)
    obj.backward()
    got = _grads(model)
    worst, total = _errors(got, ref2)
    _, yard2 = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), pos, device="cuda", loss_fn=objective,
                                               dtype=torch.bfloat16)
    yw, yt = _errors(yard2, ref2)
    # the same two numbers for stock HF run in bf16 (box["lp"] now holds ITS log-probs): what two layers of bf16
    # activations cost any implementation at this size. The kernel-level bound (2e-3 * scale on identical hidden
    # states) is tests/test_gpu_rl_logprobs.py; here the model-level error must be within 2.5e-3 relative (Frobenius; measured 2.0e-3, stock HF-bf16: 3.1e-3) and
    # its worst single token no worse than stock HF-bf16's worst token (x1.25) or 2e-3 * scale, whichever is larger.
    yerr = (box["lp"] - oracle_lp).abs().max().item()
    yrel = ((box["lp"] - oracle_lp).norm() / oracle_lp.norm()).item()
    _report("config5_logprobs", max_abs_err=err, rel_fro=rel, hf_bf16_max_abs_err=yerr, hf_bf16_rel_fro=yrel, scale=scale,
            objective=float(obj), oracle_objective=float(ref_obj),
            worst_grad_rel_fro=worst, total_grad_rel_fro=total, hf_bf16_worst_grad_rel_fro=yw, hf_bf16_total_grad_rel_fro=yt)
    assert rel <= 2.5e-3 and rel <= yrel, (rel, yrel)
    assert err <= max(2e-3 * scale, 1.25 * yerr), (err, scale, yerr)
    assert abs(float(obj) - float(ref_obj)) <= 2e-3 * max(1.0, abs(float(ref_obj)))
    assert worst < min(2 * WORST_TOL, max(WORST_TOL, 1.25 * yw)) and total < min(2 * TOTAL_TOL, max(TOTAL_TOL, 1.25 * yt)), \
        (worst, total, yw, yt)


# ------------------------------------------------------------------------------------------------------------------
def test_config4_qwen2_vl_7b_tower_nf4_lora_r32_seq4096_mrope():
    from transformers import Qwen2VLConfig
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.kernels import attention as flash
    from unsloth_amd.models import fast_layer
    vocab = 152064
    vl = Qwen2VLConfig(text_config=dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=2, num_attention_heads=28,
                                        num_key_value_heads=4, vocab_size=vocab, max_position_embeddings=32768, rms_norm_eps=1e-6,
                                        rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
                                        tie_word_embeddings=False),
                       vision_config=dict(depth=1, embed_dim=32, hidden_size=3584, num_heads=2))
    model = _build(vl, r=32, max_seq=4096, fast_model=True)
    base = model.get_base_model()
    assert type(base).__name__ == "Qwen2ForCausalLM"
    assert base.model.layers[0].self_attn.q_proj.base_layer.bias is not None            # Qwen2: q/k/v carry a bias ...
    g = torch.Generator().manual_seed(4)
    for layer in base.model.layers:                                                     # ... (random-init gives zeros)
        for n in ("q_proj", "k_proj", "v_proj"):
            b = getattr(layer.self_attn, n).base_layer.bias
            b.data.copy_((torch.randn(b.shape, generator=g) * 0.1).to(b.device, b.dtype))
    assert base._unsloth_amd_patched == (2, 2, 2), "biased q/k/v must stay on the grouped fused path"
    B, T = 1, 4096
    ids = torch.randint(0, vocab, (B, T), generator=g)
    # an image-like block in the middle: the temporal stream stalls while height / width run over a 32 x 32 grid
    t = torch.arange(T)
    img0, side = 1024, 32
    in_img = (t >= img0) & (t < img0 + side * side)
    k = (t - img0).clamp(min=0)
    pos_t = torch.where(in_img, torch.full_like(t, img0), torch.where(t < img0, t, t - side * side + side))
    pos_h = torch.where(in_img, img0 + k // side, pos_t)
    pos_w = torch.where(in_img, img0 + k % side, pos_t)
    pos3 = torch.stack([pos_t, pos_h, pos_w]).unsqueeze(1).contiguous()                 # [3, B, T]
    labels = ids.clone()
    labels[0, img0:img0 + side * side] = -100                                           # no loss on image tokens
    ref_loss, ref = hf_reference_loss_and_lora_grads(model, ids, labels, pos3, device="cuda")
    torch.cuda.empty_cache()
    _, yard = hf_reference_loss_and_lora_grads(model, ids, labels, pos3, device="cuda", dtype=torch.bfloat16)
    torch.cuda.empty_cache()
    calls = {"attn": [], "layer_fn": 0}
    real_native, real_layer = flash._forward_native, fast_layer.decoder_layer_forward
    flash._forward_native = lambda q, k_, v, s, band: (calls["attn"].append(q.shape[2] // k_.shape[2]), real_native(q, k_, v, s, band))[1]

    def counted(*a, **kw):
        calls["layer_fn"] += 1
        return real_layer(*a, **kw)
    import unsloth_amd.models.llama as L
    L._fast_layer.decoder_layer_forward = counted
    try:
        for mode in (False, "unsloth"):
            FastLanguageModel.for_training(model, use_gradient_checkpointing=mode)
            _zero(model)
            out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos3.to(DEV))
            out.loss.backward()
            _compare(f"config4_gc_{mode}", out.loss, _grads(model), ref_loss, ref, yard)
    finally:
        flash._forward_native = real_native
        L._fast_layer.decoder_layer_forward = real_layer
    assert calls["attn"] and all(g_ == 7 for g_ in calls["attn"])        # 7 query heads per KV head: native since round 6 (no padded copies)
    assert calls["layer_fn"] == 2, "multimodal positions must go through the whole-layer Function under 'unsloth'"


def test_config4_qwen2_vl_7b_pixel_values_to_loss_real_widths_lora_on_both_towers():
    """BASELINE config 4 END TO END at Qwen2-VL-7B's widths (ViT 1280 / 16 heads / patch 14 / merge 2 -> 3584; language
    3584 / 18944 / 28:4 / vocab 152064, NF4), 2 + 2 layers, ONE 896 x 896 image (4096 patches -> 1024 merged tokens) inside a
    4096-token sequence, LoRA r=32 on both towers: pixel_values -> patch-embed -> ViT (HIP LayerNorm, LoRA_W linears with
    bias) -> merger -> scatter -> get_rope_index positions -> fused language tower -> loss, against transformers' own
    Qwen2VLForConditionalGeneration in fp32 with the same (oracle-decoded, LoRA-merged) weights."""
    from transformers import Qwen2VLConfig
    from oracle.ref_model import hf_vl_reference_loss_and_lora_grads
    from unsloth_amd import FastVisionModel
    vocab = 152064
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=2, num_attention_heads=28,
                         num_key_value_heads=4, vocab_size=vocab, max_position_embeddings=32768, rms_norm_eps=1e-6,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
                         tie_word_embeddings=False),
        vision_config=dict(depth=2, embed_dim=1280, hidden_size=3584, num_heads=16, mlp_ratio=4, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653)
    model, _ = FastVisionModel.from_pretrained(config=cfg, max_seq_length=4096, load_in_4bit=True, device="cuda",
                                               use_gradient_checkpointing=False)
    model = FastVisionModel.get_peft_model(model, r=32, lora_alpha=32, use_gradient_checkpointing=False)
    g = torch.Generator().manual_seed(44)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
    for p in model.visual.parameters():              # random-init leaves LayerNorm at (1, 0) and biases at 0: give them values
        if p.dim() == 1 and not p.requires_grad:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.1 + (1.0 if p.mean() > 0.5 else 0.0)).to(p.device, p.dtype))
    grid = (1, 64, 64)
    n_img = grid[0] * (grid[1] // 2) * (grid[2] // 2)                     # 1024 placeholder tokens
    T, pre = 4096, 700
    row = torch.cat([torch.randint(0, 150000, (pre,), generator=g), torch.tensor([cfg.vision_start_token_id]),
                     torch.full((n_img,), cfg.image_token_id), torch.tensor([cfg.vision_end_token_id]),
                     torch.randint(0, 150000, (T - pre - n_img - 2,), generator=g)])
    ids = row.unsqueeze(0)
    mask = torch.ones_like(ids)
    thw = torch.tensor([grid])
    pix = torch.randn(grid[0] * grid[1] * grid[2], 3 * 2 * 14 * 14, generator=g)
    labels = ids.clone()
    labels[ids == cfg.image_token_id] = -100
    ref_loss, ref = hf_vl_reference_loss_and_lora_grads(model, ids, mask, pix, thw, labels, device="cuda")
    torch.cuda.empty_cache()
    _, yard = hf_vl_reference_loss_and_lora_grads(model, ids, mask, pix, thw, labels, device="cuda", dtype=torch.bfloat16)
    torch.cuda.empty_cache()
    out = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), pixel_values=pix.cuda(), image_grid_thw=thw.cuda(),
                labels=labels.cuda())
    out.loss.backward()
    got = {}
    for n, p in model.visual.named_parameters():
        if p.requires_grad:
            got["visual." + n.replace(".default.weight", "")] = p.grad.detach().float().cpu()
    for n, p in model.language.named_parameters():
        if p.requires_grad:
            got["language.layers." + n.split(".layers.", 1)[1].replace(".default.weight", "")] = p.grad.detach().float().cpu()
    assert len([k for k in got if k.startswith("visual.")]) == 2 * 4 * 2      # A and B of qkv / proj / fc1 / fc2 in 2 blocks
    _compare("config4_pixels_real_widths", out.loss, got, ref_loss, ref, yard)
