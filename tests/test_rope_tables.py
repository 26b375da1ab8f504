"""CPU: the rotary table builder (unsloth/models/llama.py:1688-1717, 1775-1945) against
transformers' own ROPE_INIT_FUNCTIONS / LlamaRotaryEmbedding -- default, llama3 (the branch Llama-3.1-8B takes) and
linear scaling. inv_freq must be BIT-identical; the bf16 cos/sin tables must equal what HF's module returns for the
same positions."""
import math

import pytest
import torch
from transformers import LlamaConfig

from unsloth_amd.models.llama import RopeTables, compute_inv_freq

LLAMA31 = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0, "low_freq_factor": 1.0,
           "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}


def cfg(rope, head_dim=128, max_pos=131072):
    return LlamaConfig(hidden_size=32 * head_dim, num_attention_heads=32, num_key_value_heads=8, head_dim=head_dim,
                       num_hidden_layers=1, intermediate_size=256, vocab_size=64, max_position_embeddings=max_pos,
                       rope_parameters=rope)


def reference_llama3_select_of_three(inv_freq, p):
    """the reference's formulation (llama.py:1701-1717): select between kept / divided / interpolated."""
    f, lo, hi, ctx = p["factor"], p["low_freq_factor"], p["high_freq_factor"], p["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv_freq
    out = torch.where(wavelen > ctx / lo, inv_freq / f, inv_freq)
    s = (ctx / wavelen - lo) / (hi - lo)
    mid = (1 - s) * inv_freq / f + s * inv_freq
    return torch.where((wavelen >= ctx / hi) & (wavelen <= ctx / lo), mid, out)


@pytest.mark.parametrize("theta,head_dim", [(10000.0, 64), (500000.0, 128), (1000000.0, 128), (10000.0, 96)])
def test_default_inv_freq_is_hf_default(theta, head_dim):
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    c = cfg({"rope_type": "default", "rope_theta": theta}, head_dim)
    inv, scaling, div = compute_inv_freq(c)
    want, want_scaling = LlamaRotaryEmbedding.compute_default_rope_parameters(c, "cpu")
    assert torch.equal(inv, want) and scaling == want_scaling == 1.0 and div == 1.0


@pytest.mark.parametrize("p", [LLAMA31, dict(LLAMA31, factor=32.0, rope_theta=500000.0),           # 3.1 / 3.2
                               dict(LLAMA31, factor=4.0, low_freq_factor=2.0, high_freq_factor=8.0,
                                    original_max_position_embeddings=4096, rope_theta=10000.0)])
def test_llama3_band_scaling_is_bit_identical(p):
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    c = cfg(p)
    inv, scaling, div = compute_inv_freq(c)
    want, want_scaling = ROPE_INIT_FUNCTIONS["llama3"](c, "cpu")
    assert torch.equal(inv, want), (inv - want).abs().max()
    assert scaling == want_scaling and div == 1.0
    base = 1.0 / (p["rope_theta"] ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128))
    assert torch.equal(inv, reference_llama3_select_of_three(base, p))
    # all three bands are populated for the real Llama-3.1 parameters
    ratio = inv / base
    assert (ratio == 1).any() and torch.isclose(ratio, torch.tensor(1 / p["factor"])).any() and ((ratio < 1) & (ratio > 1 / p["factor"] * 1.0001)).any()


@pytest.mark.parametrize("p,hf_kind", [({"rope_type": "default", "rope_theta": 500000.0}, "default"), (LLAMA31, "llama3")])
def test_tables_equal_hf_rotary_module_in_bf16(p, hf_kind):
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    c = cfg(p, max_pos=16384)
    cos, sin = RopeTables(c).get(3000, "cpu", torch.bfloat16)
    assert cos.shape[0] >= 8192 and cos.shape[0] % 8192 == 0            # grows in 8192 steps (llama.py:1906-1914)
    mod = LlamaRotaryEmbedding(c)
    pos = torch.tensor([[0, 1, 2, 17, 2047, 2999]])
    hc, hs = mod(torch.zeros(1, 6, 8, dtype=torch.bfloat16), pos)
    assert torch.equal(cos[pos[0]], hc[0]) and torch.equal(sin[pos[0]], hs[0])


def test_linear_scaling_divides_the_positions():
    """llama.py:1917-1945: t / scaling_factor (transformers scales inv_freq instead; same angles up to fp32 rounding)."""
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    p = {"rope_type": "linear", "rope_theta": 10000.0, "factor": 3.0}
    c = cfg(p, head_dim=64, max_pos=8192)
    inv, scaling, div = compute_inv_freq(c)
    base = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.int64).float() / 64))
    assert torch.equal(inv, base) and div == 3.0 and scaling == 1.0
    cos, sin = RopeTables(c).get(100, "cpu", torch.float32)
    t = torch.arange(100, dtype=torch.int64).float() / 3.0
    fr = torch.outer(t, base)
    assert torch.equal(cos[:100], torch.cat((fr, fr), -1).cos())
    hf_inv, _ = ROPE_INIT_FUNCTIONS["linear"](c, "cpu")
    hf = torch.outer(torch.arange(100).float(), hf_inv)
    assert (cos[:100, :32] - hf.cos()).abs().max() < 2e-5


def test_unsupported_variants_raise():
    with pytest.raises(NotImplementedError):
        compute_inv_freq(cfg({"rope_type": "yarn", "rope_theta": 10000.0, "factor": 2.0}))
