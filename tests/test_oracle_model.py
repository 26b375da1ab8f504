"""CPU: self-consistency of the test oracle itself.
  * oracle/ref_model.py (stock HF forward on merged weights + derived LoRA grads) == plain torch autograd over
    an UNMERGED LoRA forward (base(x) + s * B(A(x))) of the same tiny model;
  * oracle/nf4_ref.c (C restatement) == oracle/ref_ops.py (numpy restatement), bit for bit;
  * NF4 round trip properties (size-independent): dequant(quant(x)) error bound, idempotence of
    quant(dequant(quant(x))), nested-absmax dequant formula."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ref_model_matches_unmerged_autograd():
    from transformers import AutoModelForCausalLM, LlamaConfig
    from unsloth_amd import lora
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=16, vocab_size=97, max_position_embeddings=64,
                      rope_parameters={"rope_type": "default", "rope_theta": 1e4}, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    base = AutoModelForCausalLM.from_config(cfg).float()
    peft = lora.get_peft_model(base, lora.LoraConfig(r=4, lora_alpha=8, target_modules=[
        "q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]))
    for n, p in peft.named_parameters():
        if "lora_B" in n:
            torch.nn.init.normal_(p, std=0.05)

    def plain_forward(self, x):                     # PEFT's arithmetic, no custom kernels
        ad = self.active_adapters[0]
        return self.base_layer(x) + self.scaling[ad] * self.lora_B[ad](self.lora_A[ad](x))

    old = lora.LoraLayer.forward
    lora.LoraLayer.forward = plain_forward
    try:
        ids = torch.randint(0, 97, (2, 17))
        labels = ids.clone()
        labels[1, :3] = -100
        base._unsloth_amd_fast = False
        out = base(input_ids=ids, use_cache=False)
        shift = R.shift_labels(labels)
        loss = torch.nn.functional.cross_entropy(out.logits.view(-1, 97), shift.view(-1), ignore_index=-100)
        loss.backward()
    finally:
        lora.LoraLayer.forward = old
    ref_loss, ref_grads = hf_reference_loss_and_lora_grads(peft, ids, labels)
    torch.testing.assert_close(ref_loss, loss.detach(), rtol=1e-5, atol=1e-6)
    checked = 0
    for n, p in peft.named_parameters():
        if p.requires_grad:
            key = "layers." + n.split(".layers.", 1)[1].replace(".default.weight", "")
            torch.testing.assert_close(ref_grads[key], p.grad, rtol=2e-4, atol=1e-6)
            checked += 1
    assert checked == 2 * 7 * 2


@pytest.fixture(scope="module")
def cref():
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libnf4_ref.so"))
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_nf4_c_restatement_equals_numpy(cref):
    rng = np.random.default_rng(3407)
    w = (rng.standard_normal(64 * 257) * 0.02).astype(np.float32)
    w[128:192] = 0
    packed, absmax = R.nf4_quantize_np(w, 64)
    p2 = np.zeros_like(packed)
    a2 = np.zeros_like(absmax)
    cref.nf4_ref_quant(_p(w), 64, ctypes.c_int64(w.size), _p(p2), _p(a2))
    assert np.array_equal(packed, p2) and np.array_equal(absmax, a2)
    out = np.zeros(w.size, dtype=np.float32)
    cref.nf4_ref_dequant(_p(packed), _p(absmax), None, 64, ctypes.c_int64(w.size), _p(out))
    assert np.array_equal(out, R.nf4_dequantize_np(packed, absmax, 64))
    # nested statistics
    code2 = np.sort(rng.uniform(-1, 1, 256)).astype(np.float32)
    u8 = rng.integers(0, 256, absmax.size).astype(np.uint8)
    am2 = rng.uniform(0.01, 0.1, (absmax.size + 255) // 256).astype(np.float32)
    o = np.zeros(absmax.size, dtype=np.float32)
    cref.nf4_ref_dequant_absmax(_p(u8), _p(code2), _p(am2), ctypes.c_float(0.0123), 256, ctypes.c_int64(u8.size), _p(o))
    assert np.array_equal(o, R.dequantize_absmax_np(u8, code2, am2, 0.0123, 256))


def test_nf4_round_trip_properties():
    rng = np.random.default_rng(0)
    w = (rng.standard_normal(64 * 512) * 0.02).astype(np.float32)
    packed, absmax = R.nf4_quantize_np(w, 64)
    d = R.nf4_dequantize_np(packed, absmax, 64)
    # error bound: half the largest code gap (|-1 - -0.696| / 2 = 0.152) times the block absmax
    err = np.abs(d - w).reshape(-1, 64).max(axis=1)
    assert np.all(err <= 0.1520 * absmax + 1e-9)
    # every block reproduces its absmax element exactly (code +-1.0)
    assert np.allclose(np.abs(d).reshape(-1, 64).max(axis=1), absmax, rtol=0, atol=0)
    # idempotence: quantising the dequantised tensor gives the same bytes and statistics
    p2, a2 = R.nf4_quantize_np(d, 64)
    assert np.array_equal(p2, packed) and np.array_equal(a2, absmax)
    # packing order: high nibble is the EVEN element
    assert (packed[0] >> 4) == (np.abs(R.NF4_CODE - w[0] / absmax[0]).argmin())
