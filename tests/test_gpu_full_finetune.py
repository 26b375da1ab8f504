"""-m gpu: BASELINE config 3 (Llama-3-8B full fine-tuning, bf16): the pieces only a TRAINABLE base needs.
  * uamd_gemm_tn_256 -- dW[out, in] = dY^T @ X (torch.nn.Linear's weight gradient), against torch in fp32
  * uamd_rms_layernorm_dw -- the norm weight's gradient, which the reference's kernel does not return
    (rms_layernorm.py:218-240), against torch autograd on the fp32 formula
  * the model path: from_pretrained(full_finetuning=True) -> forward / backward, every parameter's gradient against
    stock HuggingFace in fp32 (oracle/ref_model.py `all_param_grads`)."""
import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,n_out,n_in", [(512, 256, 512), (2048, 4096, 4096), (1024, 1024, 4096), (2048, 14336, 4096),
                                          (2048, 4096, 14336), (192, 264, 328), (100, 64, 72), (4096, 6144, 4096)])
def test_dense_dw_matches_torch(T, n_out, n_in, dtype):
    from unsloth_amd.kernels.utils import dense_dw
    g = torch.Generator().manual_seed(T + n_out)
    dY = (torch.randn(T, n_out, generator=g) * 0.1).to(dtype).to(DEV)
    X = (torch.randn(T, n_in, generator=g) * 0.5).to(dtype).to(DEV)
    ref = dY.float().t() @ X.float()
    got = dense_dw(dY, X)
    assert got.shape == (n_out, n_in) and got.dtype == dtype
    assert rel_fro(got.float(), ref) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    # bit-identical to rounding the fp32 product once, up to accumulation order: compare against torch's own GEMM too
    lib = (dY.t() @ X).float()
    assert rel_fro(got.float(), ref) <= 1.5 * rel_fro(lib, ref) + 1e-6
    # accumulate: a second micro-batch adds onto the first
    acc = got.clone()
    dense_dw(dY, X, out=acc, accumulate=True)
    assert rel_fro(acc.float(), 2 * ref) < (6e-3 if dtype == torch.bfloat16 else 1e-3)


def test_dense_dw_adjacent_column_blocks_are_one_launch():
    """dQ | dK | dV side by side (the attention backward's layout) -> stacked [Wq; Wk; Wv] gradient in one GEMM."""
    from unsloth_amd.kernels.utils import dense_dw
    g = torch.Generator().manual_seed(3)
    T, H = 1024, 1024
    dQKV = (torch.randn(T, 1024 + 256 + 256, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    X = (torch.randn(T, H, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    got = dense_dw(dQKV, X)
    for lo, hi in ((0, 1024), (1024, 1280), (1280, 1536)):
        one = dense_dw(dQKV[:, lo:hi], X)
        assert torch.equal(got[lo:hi], one)


@pytest.mark.parametrize("wdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,dim", [(8192, 4096), (300, 2048), (7, 512), (4096, 3584)])
def test_rms_layernorm_weight_gradient(rows, dim, wdtype):
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    g = torch.Generator().manual_seed(rows)
    X = torch.randn(rows, dim, generator=g).to(torch.bfloat16).to(DEV).requires_grad_(True)
    W = (1 + 0.1 * torch.randn(dim, generator=g)).to(wdtype).to(DEV).requires_grad_(True)
    dY = (torch.randn(rows, dim, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    Y = Fast_RMS_Layernorm.apply(X, W, 1e-5, False)
    Y.backward(dY.clone())
    X32 = X.detach().float().requires_grad_(True)
    W32 = W.detach().float().requires_grad_(True)
    Y32 = W32 * (X32 * torch.rsqrt(X32.pow(2).mean(-1, keepdim=True) + 1e-5))
    Y32.backward(dY.float())
    assert W.grad is not None and W.grad.dtype == wdtype
    assert rel_fro(W.grad.float(), W32.grad) < (6e-3 if wdtype == torch.bfloat16 else 1e-4)
    assert rel_fro(X.grad.float(), X32.grad) < 1e-2
    # deterministic: the two-stage column reduction has a fixed order
    X.grad = W.grad = None
    first = None
    for _ in range(2):
        Y = Fast_RMS_Layernorm.apply(X, W, 1e-5, False)
        Y.backward(dY.clone())
        first = W.grad.clone() if first is None else first
        assert torch.equal(W.grad, first)
        X.grad = W.grad = None


def _llama_cfg(hidden, inter, heads, kv, vocab, layers, tie=False):
    from transformers import LlamaConfig
    return LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                       num_key_value_heads=kv, head_dim=128, vocab_size=vocab, rms_norm_eps=1e-5, max_position_embeddings=4096,
                       rope_parameters={"rope_type": "default", "rope_theta": 5e5}, tie_word_embeddings=tie)


def _full_model(cfg, max_seq):
    from unsloth_amd import FastLanguageModel
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=max_seq, full_finetuning=True, load_in_4bit=True,
                                                 device=DEV, random_state=3407, use_gradient_checkpointing=False)
    assert model._unsloth_full_finetuning and all(p.requires_grad for p in model.parameters())
    assert model._unsloth_amd_patched == (cfg.num_hidden_layers,) * 3, "dense blocks not installed on every layer"
    assert FastLanguageModel.get_peft_model(model, r=16) is model            # vision.py:1884-1889: no effect
    return model


def _grad_errors(got, ref):
    keys = sorted(ref)
    assert set(got) == set(keys), set(got) ^ set(keys)
    worst = max((rel_fro(got[k], ref[k]), k) for k in keys)
    total = rel_fro(torch.cat([got[k].flatten() for k in keys]), torch.cat([ref[k].flatten() for k in keys]))
    return worst, total


@pytest.mark.parametrize("tie", [False, True])
@pytest.mark.parametrize("with_buckets", [False, True])
def test_full_finetune_every_gradient_against_hf_fp32(with_buckets, tie):
    """Small widths (fast): loss and EVERY parameter's gradient -- projections (uamd_gemm_tn_256), norms
    (uamd_rms_layernorm_dw), lm_head (chunked, inside the fused linear-CE), embeddings (torch) -- against stock HF fp32;
    with and without the flat gradient buckets (direct sink writes vs autograd-returned gradients)."""
    from oracle.ref_model import hf_reference_loss_and_all_grads
    from unsloth_amd.full_finetune import FullGradBuckets
    cfg = _llama_cfg(512, 1024, 4, 2, 2048, 2, tie)
    model = _full_model(cfg, 512)
    T = 384
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 2048, (2, T), generator=g)
    labels = ids.clone()
    labels[1, :17] = -100
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0).expand(2, T).contiguous()
    ref_loss, ref = hf_reference_loss_and_all_grads(model, ids, labels, pos)
    _, yard = hf_reference_loss_and_all_grads(model, ids, labels, pos, dtype=torch.bfloat16)
    buckets = FullGradBuckets(model) if with_buckets else None
    try:
        for rep in range(2):                                  # second pass: zero_grad() without fills, same numbers
            out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
            out.loss.backward()
            if buckets is not None:
                buckets.finish()
                base = {id(b["flat_g"]): (b["flat_g"].data_ptr(), b["numel"]) for b in buckets.buckets}
                assert all(any(lo <= p.grad.data_ptr() < lo + 2 * n for lo, n in base.values()) for p in model.parameters())
            got = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
            assert abs(float(out.loss) - float(ref_loss)) <= 1e-3 * abs(float(ref_loss))
            (worst, wk), total = _grad_errors(got, ref)
            (yw, _), yt = _grad_errors(yard, ref)
            assert worst < max(2.5e-2, 1.25 * yw) and total < max(1.5e-2, 1.25 * yt), (worst, wk, total, yw, yt)
            if buckets is not None:
                buckets.zero_grad()
            else:
                for p in model.parameters():
                    p.grad = None
    finally:
        if buckets is not None:
            buckets.close()


def test_config3_llama3_8b_widths_full_finetune_seq2048_step():
    """BASELINE config 3 at its stated widths and sequence length (2 layers, batch 1 x 2048): gradients against HF fp32 on
    the GPU, then ONE sharded-AdamW step (world size 1: the shard is the whole bucket) against torch.optim.AdamW on fp32
    masters fed the same gradients."""
    from oracle.ref_model import hf_reference_loss_and_all_grads
    from unsloth_amd.full_finetune import ShardedAdamW
    cfg = _llama_cfg(4096, 14336, 32, 8, 128256, 2)
    model = _full_model(cfg, 2048)
    T = 2048
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 128256, (1, T), generator=g)
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0)
    ref_loss, ref = hf_reference_loss_and_all_grads(model, ids, ids.clone(), pos, device="cuda")
    torch.cuda.empty_cache()
    _, yard = hf_reference_loss_and_all_grads(model, ids, ids.clone(), pos, device="cuda", dtype=torch.bfloat16)
    torch.cuda.empty_cache()
    opt = ShardedAdamW(model, lr=1e-3, weight_decay=0.01)
    masters = {n: p.detach().float().clone() for n, p in model.named_parameters()}
    out = model(input_ids=ids.to(DEV), labels=ids.to(DEV), position_ids=pos.to(DEV))
    out.loss.backward()
    opt.buckets.finish()
    got = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
    assert abs(float(out.loss) - float(ref_loss)) <= 1e-3 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    (worst, wk), total = _grad_errors(got, ref)
    (yw, _), yt = _grad_errors(yard, ref)
    import json, os
    rec = dict(loss=float(out.loss), oracle_loss=float(ref_loss), worst_grad_rel_fro=worst, worst_param=wk,
               total_grad_rel_fro=total, hf_bf16_worst_grad_rel_fro=yw, hf_bf16_total_grad_rel_fro=yt)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rec, open("gpurun_out/config3_parity.json", "w"), indent=1)
    assert worst < min(5e-2, max(2.5e-2, 1.25 * yw)) and total < min(3e-2, max(1.5e-2, 1.25 * yt)), rec
    # one optimizer step
    grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
    opt.step()
    worst_p = 0.0
    for n, p in model.named_parameters():
        m32 = torch.nn.Parameter(masters[n])
        m32.grad = grads[n]
        # HF Trainer's rule (get_decay_parameter_names): no decay on biases and norm weights, i.e. on 1-D parameters
        torch.optim.AdamW([m32], lr=1e-3, weight_decay=0.01 if m32.dim() > 1 else 0.0).step()
        assert torch.equal(p.detach(), m32.detach().to(torch.bfloat16)) or \
            (p.detach().float() - m32.detach().to(torch.bfloat16).float()).abs().max() <= 2 ** -8 * m32.abs().max(), n
        worst_p = max(worst_p, (p.detach().float() - m32.detach()).abs().max().item())
    opt.zero_grad()
    assert all(p.grad is None for p in model.parameters())
    opt.buckets.close()


@pytest.mark.parametrize("arch,gc", [("qwen2", False), ("llama", True), ("qwen2", True)])
def test_full_finetune_biased_projections_and_reentrant_checkpointing(arch, gc):
    """Qwen2's q/k/v BIASES train too under full fine-tuning (column sums of dQ / dK / dV), and torch's reentrant per-layer
    checkpointing (use_gradient_checkpointing=True: the reference's default for full fine-tuning through HF Trainer) re-runs
    the dense blocks inside the backward with the gradient sinks attached."""
    from oracle.ref_model import hf_reference_loss_and_all_grads
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.full_finetune import FullGradBuckets
    if arch == "qwen2":
        from transformers import Qwen2Config
        cfg = Qwen2Config(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, vocab_size=2048, rms_norm_eps=1e-6, max_position_embeddings=1024,
                          rope_parameters={"rope_type": "default", "rope_theta": 1e6}, tie_word_embeddings=False)
    else:
        cfg = _llama_cfg(512, 1024, 4, 2, 2048, 2)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=512, full_finetuning=True, device=DEV,
                                                 random_state=3407, use_gradient_checkpointing=gc)
    assert model._unsloth_amd_patched == (2, 2, 2)
    g = torch.Generator().manual_seed(9)
    if arch == "qwen2":
        for layer in model.model.layers:
            for n in ("q_proj", "k_proj", "v_proj"):
                b = getattr(layer.self_attn, n).bias
                assert b is not None and b.requires_grad
                b.data.copy_((torch.randn(b.shape, generator=g) * 0.1).to(b.device, b.dtype))
    T = 320
    ids = torch.randint(0, 2048, (2, T), generator=g)
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0).expand(2, T).contiguous()
    ref_loss, ref = hf_reference_loss_and_all_grads(model, ids, ids.clone(), pos)
    _, yard = hf_reference_loss_and_all_grads(model, ids, ids.clone(), pos, dtype=torch.bfloat16)
    buckets = FullGradBuckets(model)
    try:
        out = model(input_ids=ids.to(DEV), labels=ids.to(DEV), position_ids=pos.to(DEV))
        out.loss.backward()
        buckets.finish()
        got = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
        assert abs(float(out.loss) - float(ref_loss)) <= 1e-3 * abs(float(ref_loss))
        (worst, wk), total = _grad_errors(got, ref)
        (yw, _), yt = _grad_errors(yard, ref)
        assert worst < max(2.5e-2, 1.25 * yw) and total < max(1.5e-2, 1.25 * yt), (worst, wk, total, yw, yt)
    finally:
        buckets.close()
