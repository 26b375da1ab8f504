"""FastVisionModel (Qwen2-VL; BASELINE config 4).
CPU: the multimodal position ids (integer work) bit-exact against transformers' Qwen2VLModel.get_rope_index.
GPU: pixel_values -> loss through the vision tower (LayerNorm kernel, LoRA'd linears) + the fused language tower against
transformers' Qwen2VLForConditionalGeneration in fp32 with the same weights."""
import pytest
import torch


def _vl_config(hidden=256, heads=4, kv=2, inter=512, vocab=1200, layers=2, depth=2, embed=128):
    from transformers import Qwen2VLConfig
    return Qwen2VLConfig(
        text_config=dict(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                         num_key_value_heads=kv, vocab_size=vocab, max_position_embeddings=1024, rms_norm_eps=1e-6,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [hidden // heads // 8, 3 * hidden // heads // 16, 3 * hidden // heads // 16]},
                         tie_word_embeddings=False),
        vision_config=dict(depth=depth, embed_dim=embed, hidden_size=hidden, num_heads=4, mlp_ratio=2, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=1100, video_token_id=1101, vision_start_token_id=1102, vision_end_token_id=1103)


def _sample(cfg, grids, B=2, gen=None):
    """token rows with one image each: [text | <vision_start> image placeholders <vision_end> | text], right-padded."""
    m = cfg.vision_config.spatial_merge_size
    rows, lens = [], []
    for (t, h, w), pre, post in zip(grids, (5, 11), (9, 4)):
        n = t * (h // m) * (w // m)
        row = torch.cat([torch.randint(0, 1000, (pre,), generator=gen), torch.tensor([cfg.vision_start_token_id]),
                         torch.full((n,), cfg.image_token_id), torch.tensor([cfg.vision_end_token_id]),
                         torch.randint(0, 1000, (post,), generator=gen)])
        rows.append(row)
        lens.append(len(row))
    T = max(lens)
    ids = torch.zeros(B, T, dtype=torch.long)
    mask = torch.zeros(B, T, dtype=torch.long)
    for b, row in enumerate(rows):
        ids[b, :len(row)] = row
        mask[b, :len(row)] = 1
    return ids, mask


def test_mrope_position_ids_match_transformers_get_rope_index():
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLModel
    from unsloth_amd.models.vision import mrope_position_ids
    cfg = _vl_config()
    hf = Qwen2VLModel.__new__(Qwen2VLModel)          # only the config is read by get_rope_index
    torch.nn.Module.__init__(hf)
    hf.config = cfg
    gen = torch.Generator().manual_seed(0)
    for grids in ([(1, 4, 6), (1, 8, 4)], [(2, 4, 4), (1, 2, 10)]):
        ids, mask = _sample(cfg, grids, gen=gen)
        thw = torch.tensor(grids)
        types = (ids == cfg.image_token_id).int()
        for am in (mask, None):
            if am is None and int(mask.min()) == 0:
                # without a mask the padding zeros are ordinary text tokens for both implementations
                pass
            want, want_delta = hf.get_rope_index(ids, mm_token_type_ids=types, image_grid_thw=thw, attention_mask=am)
            got, delta = mrope_position_ids(ids, thw, None, am, cfg.image_token_id, cfg.video_token_id,
                                            cfg.vision_config.spatial_merge_size)
            assert torch.equal(got, want) and torch.equal(delta.view(-1), want_delta.view(-1))
    # text only: three identical aranges
    ids = torch.randint(0, 1000, (2, 17), generator=gen)
    got, delta = mrope_position_ids(ids, None, None, None, cfg.image_token_id, cfg.video_token_id, 2)
    assert torch.equal(got, torch.arange(17).expand(3, 2, 17)) and int(delta.abs().sum()) == 0


@pytest.mark.gpu
def test_fast_vision_model_pixel_values_to_loss_matches_hf_fp32():
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLForConditionalGeneration
    from unsloth_amd import FastVisionModel
    from unsloth_amd.kernels import layernorm as LN
    from unsloth_amd.kernels.rms_layernorm import patch_rms_layernorm, unpatch_rms_layernorm
    cfg = _vl_config()
    model, _ = FastVisionModel.from_pretrained(config=cfg, max_seq_length=256, load_in_4bit=False, device="cuda",
                                               use_gradient_checkpointing=False)
    gen = torch.Generator().manual_seed(3)
    for p in model.visual.parameters():              # random-init leaves LayerNorm at (1, 0) and biases at 0: give them values
        if p.dim() == 1:
            p.data.copy_((torch.randn(p.shape, generator=gen) * 0.1 + (1.0 if p.mean() > 0.5 else 0.0)).to(p.device, p.dtype))
    grids = [(1, 4, 6), (1, 8, 4)]
    ids, mask = _sample(cfg, grids, gen=gen)
    thw = torch.tensor(grids)
    n_patches = int(sum(t * h * w for t, h, w in grids))
    pix = torch.randn(n_patches, 3 * 2 * 14 * 14, generator=gen)
    labels = ids.clone()
    labels[mask == 0] = -100
    labels[ids == cfg.image_token_id] = -100
    calls = []
    real = LN.Fast_Layernorm.apply
    LN.Fast_Layernorm.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
    try:
        out = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), pixel_values=pix.cuda(), image_grid_thw=thw.cuda(),
                    labels=labels.cuda())
    finally:
        LN.Fast_Layernorm.apply = real
    assert len(calls) >= 2 * cfg.vision_config.depth + 1, "the ViT's LayerNorms must run through the HIP kernel"
    # the same weights in transformers' own model, fp32, host
    unpatch_rms_layernorm()
    LN.unpatch_layernorm()
    try:
        hf = Qwen2VLForConditionalGeneration(cfg).float().eval()
        sd = {}
        for k, v in model.visual.state_dict().items():
            sd["model.visual." + k] = v.float().cpu()
        for k, v in model.language.state_dict().items():
            if k.startswith("model."):
                sd["model.language_model." + k[len("model."):]] = v.float().cpu()
            else:
                sd[k] = v.float().cpu()
        missing, unexpected = hf.load_state_dict(sd, strict=False)
        assert not [m for m in missing if "rotary" not in m and "inv_freq" not in m], missing
        assert not unexpected, unexpected
        with torch.no_grad():
            # (no `labels=`: transformers' loss mapping is patched to the product's HIP cross entropy; the reference loss
            #  is torch's own on the fp32 logits)
            logits = hf(input_ids=ids, attention_mask=mask, pixel_values=pix, image_grid_thw=thw,
                        mm_token_type_ids=(ids == cfg.image_token_id).int()).logits.float()
            ref_loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1),
                                                         ignore_index=-100)
    finally:
        patch_rms_layernorm()
        LN.patch_layernorm()
    assert abs(float(out.loss) - float(ref_loss)) <= 2e-3 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    # the image reaches the loss: different pixels, different loss
    out2 = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), pixel_values=(pix * 0.5).cuda(),
                 image_grid_thw=thw.cuda(), labels=labels.cuda())
    assert float(out2.loss) != float(out.loss)


@pytest.mark.gpu
def test_fast_vision_model_lora_on_both_towers_trains():
    """get_peft_model(finetune_vision_layers, finetune_language_layers): LoRA factors on the ViT linears (through LoRA_W)
    and on the language tower (fused hooks incl. biased q/k/v), gradients everywhere, loss goes down."""
    from unsloth_amd import FastVisionModel
    from unsloth_amd import lora as L
    cfg = _vl_config()
    model, _ = FastVisionModel.from_pretrained(config=cfg, max_seq_length=256, load_in_4bit=True, device="cuda",
                                               use_gradient_checkpointing=False)
    model = FastVisionModel.get_peft_model(model, r=8, lora_alpha=8, use_gradient_checkpointing="unsloth")
    vis = [n for n, m in model.visual.named_modules() if isinstance(m, L.LoraLayer)]
    assert len(vis) == 4 * cfg.vision_config.depth, vis
    assert model.language.get_base_model()._unsloth_amd_patched == (2, 2, 2)
    gen = torch.Generator().manual_seed(4)
    grids = [(1, 4, 6), (1, 8, 4)]
    ids, mask = _sample(cfg, grids, gen=gen)
    thw = torch.tensor(grids)
    pix = torch.randn(int(sum(t * h * w for t, h, w in grids)), 3 * 2 * 14 * 14, generator=gen)
    labels = ids.clone()
    labels[mask == 0] = -100
    labels[ids == cfg.image_token_id] = -100
    params = [p for p in model.parameters() if p.requires_grad]
    assert all("lora_" in n for n, p in model.named_parameters() if p.requires_grad)
    opt = torch.optim.AdamW(params, lr=2e-3)
    batch = dict(input_ids=ids.cuda(), attention_mask=mask.cuda(), pixel_values=pix.cuda(), image_grid_thw=thw.cuda(),
                 labels=labels.cuda())
    losses = []
    for _ in range(12):
        loss = model(**batch).loss
        loss.backward()
        if not losses:
            vis_g = [p.grad for n, p in model.visual.named_parameters() if "lora_B" in n]
            lang_g = [p.grad for n, p in model.language.named_parameters() if "lora_B" in n]
            assert vis_g and all(g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0 for g in vis_g)
            assert lang_g and all(g is not None and float(g.abs().sum()) > 0 for g in lang_g)
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.05, losses


@pytest.mark.gpu
def test_fast_vision_model_lora_gradients_on_both_towers_match_hf_fp32():
    """pixel_values -> loss -> LoRA gradients of the ViT linears (LoRA_W with bias) and of the fused language tower, against
    transformers' Qwen2VLForConditionalGeneration in fp32 with merged weights (oracle/ref_model.py)."""
    from oracle.ref_model import hf_vl_reference_loss_and_lora_grads
    from tests._util import rel_fro
    from unsloth_amd import FastVisionModel
    cfg = _vl_config()
    model, _ = FastVisionModel.from_pretrained(config=cfg, max_seq_length=256, load_in_4bit=True, device="cuda",
                                               use_gradient_checkpointing=False)
    model = FastVisionModel.get_peft_model(model, r=8, lora_alpha=8, use_gradient_checkpointing=False)
    gen = torch.Generator().manual_seed(6)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=gen) * 0.05).to(p.device))
    grids = [(1, 4, 6), (1, 8, 4)]
    ids, mask = _sample(cfg, grids, gen=gen)
    thw = torch.tensor(grids)
    pix = torch.randn(int(sum(t * h * w for t, h, w in grids)), 3 * 2 * 14 * 14, generator=gen)
    labels = ids.clone()
    labels[mask == 0] = -100
    labels[ids == cfg.image_token_id] = -100
    ref_loss, ref = hf_vl_reference_loss_and_lora_grads(model, ids, mask, pix, thw, labels)
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
    assert set(got) == set(ref)
    assert abs(float(out.loss) - float(ref_loss)) <= 2e-3 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    worst = max((rel_fro(got[k], ref[k]), k) for k in got)
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]), torch.cat([ref[k].flatten() for k in sorted(got)]))
    assert worst[0] < 4e-2 and total < 2e-2, (worst, total)
