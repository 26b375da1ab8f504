"""-m gpu: the BASELINE.json configurations as parity cases AT THEIR REAL WIDTHS (few layers, synthetic weights): the
whole drop-in surface (from_pretrained -> get_peft_model -> forward / backward) on the HIP path against the
implementation-independent oracle (stock HF model on the CPU in fp32 over oracle-dequantised weights + merged LoRA).
  config 1  TinyLlama-1.1B widths, LoRA r=8 on a 16-bit base, seq 512, batch 1 (head_dim 64: native to csrc/attention.hip since round 6)
  config 2  Llama-3-8B widths, QLoRA NF4 r=16 (the benchmark's model, one layer, 512 tokens)
  config 5  Mistral-7B widths, LoRA r=16, sliding window (band kernels) -- the fused linear-CE path of the DPO/GRPO runs
  config 4  (Qwen2-VL-7B text tower, mrope, 28:4 heads) lives in tests/test_gpu_mrope.py at its real widths
North star: loss within 1e-3 of the reference path; LoRA gradients within the bf16 end-to-end bound of test_gpu_model."""
import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _build(cfg, load_in_4bit, r, max_seq):
    from unsloth_amd import FastLanguageModel
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=max_seq, load_in_4bit=load_in_4bit, device=DEV,
                                                 random_state=3407, use_gradient_checkpointing=False)
    model = FastLanguageModel.get_peft_model(model, r=r, lora_alpha=r, use_gradient_checkpointing=False, random_state=3407)
    g = torch.Generator().manual_seed(3407)
    for n, p in model.named_parameters():
        if "lora_B" in n:                      # PEFT's default B = 0 would zero half of the gradients
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(DEV))
    return model


def _check(model, T, vocab, n_layers, loss_tol=1e-3):
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, vocab, (1, T), generator=g)
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0)
    assert model.get_base_model()._unsloth_amd_patched == (n_layers,) * 3, "fused hooks not installed on every layer"
    out = model(input_ids=ids.to(DEV), labels=ids.to(DEV), position_ids=pos.to(DEV))
    out.loss.backward()
    got = {"layers." + n.split(".layers.", 1)[1].replace(".default.weight", ""): p.grad.detach().float().cpu()
           for n, p in model.named_parameters() if p.requires_grad}
    ref_loss, ref = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), pos)
    assert abs(float(out.loss) - float(ref_loss)) <= loss_tol * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    assert set(got) == set(ref)
    worst = max(rel_fro(got[k], ref[k]) for k in got)
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]), torch.cat([ref[k].flatten() for k in sorted(got)]))
    assert worst < 2.5e-2 and total < 1.5e-2, (worst, total)


def test_config1_tinyllama_widths_lora_r8_seq512():
    from transformers import LlamaConfig
    cfg = LlamaConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=4, head_dim=64, vocab_size=32000, rms_norm_eps=1e-5, max_position_embeddings=2048,
                      rope_parameters={"rope_type": "default", "rope_theta": 1e4}, tie_word_embeddings=False)
    _check(_build(cfg, load_in_4bit=False, r=8, max_seq=512), T=512, vocab=32000, n_layers=2)


def test_config2_llama3_8b_widths_qlora_nf4_r16():
    from transformers import LlamaConfig
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32,
                      num_key_value_heads=8, head_dim=128, vocab_size=128256, rms_norm_eps=1e-5, max_position_embeddings=8192,
                      rope_parameters={"rope_type": "llama3", "rope_theta": 5e5, "factor": 8.0, "low_freq_factor": 1.0,
                                       "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
                      tie_word_embeddings=False)
    _check(_build(cfg, load_in_4bit=True, r=16, max_seq=512), T=512, vocab=128256, n_layers=1)


def test_config5_mistral_7b_widths_lora_r16_sliding_window():
    from transformers import MistralConfig
    cfg = MistralConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32,
                        num_key_value_heads=8, head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, max_position_embeddings=4096,
                        sliding_window=192, rope_parameters={"rope_type": "default", "rope_theta": 1e4},
                        tie_word_embeddings=False)
    _check(_build(cfg, load_in_4bit=True, r=16, max_seq=512), T=512, vocab=32000, n_layers=1)
