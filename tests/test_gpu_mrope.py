"""-m gpu: multimodal RoPE (Qwen2-VL text tower, BASELINE config 4 / SURVEY 8 f4) on the HIP RoPE kernel against
transformers' own `apply_multimodal_rotary_pos_emb` + `Qwen2VLRotaryEmbedding` (the implementation the reference's
VLM path ends up running). Forward and backward; bf16 results must be BIT-identical (both sides multiply and add in
the activation dtype: three roundings per element), fp32 within 2 ulp."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _hf_tables(pos3, D, theta, dtype):
    """cos/sin [3, B, T, D] exactly as Qwen2VLRotaryEmbedding.forward builds them (fp32 angles, cast at the end)."""
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    freqs = pos3[..., None].float() * inv                       # [3, B, T, D/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _table(n, D, theta, dtype):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    fr = torch.outer(torch.arange(n, dtype=torch.int64).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("D,section,Hq,Hk", [(128, (16, 24, 24), 28, 4), (64, (8, 12, 12), 4, 2), (32, (3, 5, 8), 2, 1)])
def test_mrope_matches_transformers(dtype, D, section, Hq, Hk):
    from transformers.models.qwen2_vl.modeling_qwen2_vl import apply_multimodal_rotary_pos_emb
    from unsloth_amd.kernels import fast_mrope_embedding
    B, T, theta = 2, 37, 1e6
    g = torch.Generator().manual_seed(D)
    # text tokens: all three streams equal; an image block: temporal constant, height / width a grid
    pos3 = torch.arange(T)[None, None, :].repeat(3, B, 1)
    pos3[0, :, 10:22] = 10
    pos3[1, :, 10:22] = 10 + torch.arange(12) // 4
    pos3[2, :, 10:22] = 10 + torch.arange(12) % 4
    pos3[:, 1] += 5
    Q = torch.randn(B, Hq, T, D, generator=g).to(dtype)
    K = torch.randn(B, Hk, T, D, generator=g).to(dtype)
    cos3, sin3 = _hf_tables(pos3, D, theta, dtype)
    wantQ, wantK = apply_multimodal_rotary_pos_emb(Q, K, cos3, sin3, list(section))
    cos, sin = _table(64, D, theta, dtype)
    # strided [B,T,H,D] storage viewed as [B,H,T,D], like the model's projection outputs
    Qd = Q.transpose(1, 2).contiguous().to(DEV).transpose(1, 2).requires_grad_(True)
    Kd = K.transpose(1, 2).contiguous().to(DEV).transpose(1, 2).requires_grad_(True)
    q, k = fast_mrope_embedding(Qd * 1.0, Kd * 1.0, cos.to(DEV), sin.to(DEV), pos3.to(DEV), section)
    if dtype == torch.float32:
        torch.testing.assert_close(q.cpu(), wantQ, rtol=3e-7, atol=3e-7)
        torch.testing.assert_close(k.cpu(), wantK, rtol=3e-7, atol=3e-7)
    else:
        assert torch.equal(q.cpu(), wantQ) and torch.equal(k.cpu(), wantK)
    # backward = the inverse rotation (sin -> -sin) of the upstream gradient; oracle: autograd through HF's function
    dQ = torch.randn(B, Hq, T, D, generator=g).to(dtype)
    dK = torch.randn(B, Hk, T, D, generator=g).to(dtype)
    Qr, Kr = Q.clone().requires_grad_(True), K.clone().requires_grad_(True)
    a, b = apply_multimodal_rotary_pos_emb(Qr, Kr, cos3, sin3, list(section))
    torch.autograd.backward([a, b], [dQ, dK])
    torch.autograd.backward([q, k], [dQ.to(DEV), dK.to(DEV)])
    tol = dict(rtol=3e-7, atol=3e-7) if dtype == torch.float32 else dict(rtol=0, atol=0)
    if dtype == torch.float32:
        torch.testing.assert_close(Qd.grad.cpu(), Qr.grad, **tol)
        torch.testing.assert_close(Kd.grad.cpu(), Kr.grad, **tol)
    else:
        # autograd's backward of (q*cos + rotate_half(q)*sin) rounds the two products and their sum like the kernel
        assert torch.equal(Qd.grad.cpu(), Qr.grad) and torch.equal(Kd.grad.cpu(), Kr.grad)


def test_mrope_with_equal_streams_is_ordinary_rope():
    from unsloth_amd.kernels import fast_mrope_embedding, fast_rope_embedding
    B, H, Hk, T, D = 1, 8, 2, 50, 128
    g = torch.Generator().manual_seed(1)
    Q = torch.randn(B, H, T, D, generator=g).to(torch.bfloat16).to(DEV)
    K = torch.randn(B, Hk, T, D, generator=g).to(torch.bfloat16).to(DEV)
    cos, sin = _table(128, D, 5e5, torch.bfloat16)
    idx = torch.randint(0, 128, (B * T,), generator=g).to(torch.int32)
    q1, k1 = fast_rope_embedding(Q.clone(), K.clone(), cos.to(DEV), sin.to(DEV), idx.to(DEV))
    q2, k2 = fast_mrope_embedding(Q.clone(), K.clone(), cos.to(DEV), sin.to(DEV), idx.view(1, B, T).repeat(3, 1, 1).to(DEV),
                                  (16, 24, 24))
    assert torch.equal(q1, q2) and torch.equal(k1, k2)


@pytest.mark.parametrize("hidden,heads,kv,inter,vocab", [
    (512, 4, 2, 1024, 1000), (896, 7, 1, 1024, 1000),       # 7:1 = Qwen2-VL-7B's 28:4 grouping (G = 7)
    (3584, 28, 4, 18944, 152064)])                         # BASELINE config 4: the text tower at its real widths
def test_fastmodel_language_tower_matches_hf_qwen2_vl_text_model(hidden, heads, kv, inter, vocab):
    """FastModel on a Qwen2-VL config = its language tower on the hand-kernel path: same weights in transformers'
    Qwen2VLTextModel (fp32, host) + lm_head give the same loss for [3, B, T] multimodal positions, and text-only [B, T]
    positions equal three identical streams."""
    import torch.nn.functional as F
    from transformers import Qwen2VLConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextModel
    from unsloth_amd import FastModel
    from unsloth_amd.kernels.rms_layernorm import unpatch_rms_layernorm
    vl = Qwen2VLConfig(text_config=dict(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=2 if hidden < 2048 else 1,
                                        num_attention_heads=heads, num_key_value_heads=kv, vocab_size=vocab, max_position_embeddings=512, rms_norm_eps=1e-6,
                                        rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
                                        tie_word_embeddings=False),
                       vision_config=dict(depth=1, embed_dim=32, hidden_size=512, num_heads=2))
    model, _ = FastModel.from_pretrained(config=vl, max_seq_length=256, load_in_4bit=False, device="cuda",
                                         use_gradient_checkpointing=False)
    assert type(model).__name__ == "Qwen2ForCausalLM" and model._unsloth_amd_fast
    B, T = 2, 48
    gen = torch.Generator().manual_seed(5)
    ids = torch.randint(0, vocab, (B, T), generator=gen)
    pos3 = torch.stack([torch.arange(T).expand(B, T), torch.randint(0, 30, (B, T), generator=gen),
                        torch.randint(0, 30, (B, T), generator=gen)])                      # temporal / height / width
    labels = ids.clone()
    model.train()
    from unsloth_amd.kernels import attention as flash
    calls, real = [], flash._forward_native
    flash._forward_native = lambda q, k, v, s, band: (calls.append(q.shape[2] // k.shape[2]), real(q, k, v, s, band))[1]
    try:
        out = model(input_ids=ids.cuda(), labels=labels.cuda(), position_ids=pos3.cuda())
    finally:
        flash._forward_native = real
    assert calls and all(1 <= g_ <= 8 for g_ in calls)                  # the hand kernels ran (7 query heads on a KV head: native)
    out_t = model(input_ids=ids.cuda(), labels=labels.cuda(), position_ids=torch.arange(T).expand(B, T).cuda())
    out_3 = model(input_ids=ids.cuda(), labels=labels.cuda(), position_ids=torch.arange(T).expand(3, B, T).contiguous().cuda())
    assert abs(float(out_t.loss) - float(out_3.loss)) < 1e-6
    # independent check: transformers' text model on the host in fp32 with the same (bf16-rounded) weights
    unpatch_rms_layernorm()
    ref = Qwen2VLTextModel(vl.text_config).float()
    sd = {k[len("model."):]: v.float().cpu() for k, v in model.state_dict().items() if k.startswith("model.")}
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "rotary" not in m], missing
    with torch.no_grad():
        h = ref(input_ids=ids, position_ids=pos3).last_hidden_state
        logits = h @ model.lm_head.weight.float().cpu().t()
        want = F.cross_entropy(logits[:, :-1].reshape(-1, vocab), labels[:, 1:].reshape(-1))
    assert abs(float(out.loss) - float(want)) < 2e-3 * abs(float(want)), (float(out.loss), float(want))
    assert float(out.loss) != float(out_t.loss)                       # the height / width streams reach the result
