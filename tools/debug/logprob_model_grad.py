"""Bisect of the config-5 log-prob leg at model level: LoRA gradients of a weighted per-token log-prob objective through
UNSLOTH_RETURN_HIDDEN_STATES + chunked_hidden_states_selective_log_softmax against the fp32 HF oracle."""
import os
import sys
import torch
sys.path.insert(0, ".")
from transformers import MistralConfig
from oracle.ref_model import hf_reference_loss_and_lora_grads
from unsloth_amd import FastLanguageModel
from unsloth_amd.models.rl_replacements import chunked_hidden_states_selective_log_softmax

DEV = torch.device("cuda", 0)


def run(T, layers, ce_first, hidden=4096, inter=14336, V=32000, chunks=4, half_mask=True):
    cfg = MistralConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=hidden // 128,
                        num_key_value_heads=max(1, hidden // 512), head_dim=128, vocab_size=V, rms_norm_eps=1e-5,
                        max_position_embeddings=32768, sliding_window=4096,
                        rope_parameters={"rope_type": "default", "rope_theta": 1e4}, tie_word_embeddings=False)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=T, load_in_4bit=True, device=DEV, random_state=3407,
                                                 use_gradient_checkpointing=False)
    model = FastLanguageModel.get_peft_model(model, r=16, lora_alpha=16, use_gradient_checkpointing=False, random_state=3407)
    g = torch.Generator().manual_seed(3407)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(DEV))
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (1, T), generator=g)
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0)
    if ce_first:
        model(input_ids=ids.to(DEV), labels=ids.to(DEV), position_ids=pos.to(DEV)).loss.backward()
        for p in model.parameters():
            p.grad = None
    mask = torch.zeros(1, T - 1)
    mask[0, (T // 2 if half_mask else 0):] = 1.0
    wts = torch.randn(1, T - 1, generator=g) * mask
    nxt = ids[:, 1:]

    def objective(logits):
        lp = torch.log_softmax(logits[:, :-1].float(), dim=-1).gather(-1, nxt.to(logits.device).unsqueeze(-1)).squeeze(-1)
        return -(lp * wts.to(lp.device)).sum() / mask.sum()

    ref_obj, ref = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), pos, device="cuda", loss_fn=objective)
    os.environ["UNSLOTH_RETURN_HIDDEN_STATES"] = "1"
    hidden = model(input_ids=ids.to(DEV), position_ids=pos.to(DEV)).logits
    os.environ["UNSLOTH_RETURN_HIDDEN_STATES"] = "0"
    hidden.retain_grad()
    lm_head = model.get_base_model().lm_head.weight
    lp = chunked_hidden_states_selective_log_softmax(hidden[:, :-1], lm_head, nxt.to(DEV), chunks=chunks)
    obj = -(lp * wts.to(DEV)).sum() / mask.sum().to(DEV)
    obj.backward()
    got = {"layers." + n.split(".layers.", 1)[1].replace(".default.weight", ""): p.grad.detach().float().cpu()
           for n, p in model.named_parameters() if p.requires_grad}
    # the same objective through materialised logits + torch autograd on the PRODUCT's hidden states: is d(hidden) right?
    h32 = hidden.detach().float().requires_grad_(True)
    lg = h32[0, :-1] @ lm_head.float().t()
    lp2 = torch.log_softmax(lg, -1).gather(-1, nxt.to(DEV)[0].unsqueeze(-1)).squeeze(-1)
    (-(lp2 * wts.to(DEV)[0]).sum() / mask.sum().to(DEV)).backward()
    dh_err = ((hidden.grad.float() - h32.grad).norm() / h32.grad.norm()).item()
    errs = {k: ((got[k] - ref[k]).norm() / ref[k].norm()).item() for k in sorted(got)}
    cos = {k: (torch.dot(got[k].flatten(), ref[k].flatten()) / (got[k].norm() * ref[k].norm())).item() for k in sorted(got)}
    tot = (torch.cat([got[k].flatten() - ref[k].flatten() for k in sorted(got)]).norm()
           / torch.cat([ref[k].flatten() for k in sorted(got)]).norm()).item()
    worst = max(errs, key=errs.get)
    print(f"T={T} layers={layers} ce_first={ce_first} hidden={hidden} chunks={chunks} half_mask={half_mask}: obj {float(obj):.5f} / {float(ref_obj):.5f} "
          f"d(hidden) rel {dh_err:.4f}  total {tot:.4f}  worst {worst} {errs[worst]:.4f} cos {cos[worst]:.3f}", flush=True)
    if tot > 0.1:
        for k in sorted(errs):
            print(f"    {k:45s} rel {errs[k]:.3f} cos {cos[k]:.3f} |got| {got[k].norm():.3e} |ref| {ref[k].norm():.3e}")
    del model
    torch.cuda.empty_cache()


run(512, 1, False, hidden=1024, inter=2048, V=4096)
run(512, 1, False)
run(4096, 1, False)
run(4096, 2, False)
run(4096, 2, True)
run(4096, 2, False, half_mask=False)
