"""bitsandbytes-4bit safetensors layout: write -> read round trip is BYTE exact, key names are the ones transformers'
bnb quantizer lists (quantizers/quantizer_bnb_4bit.py get_weight_conversions), the quant-state json carries the
fields bitsandbytes QuantState.from_dict reads. CPU only: packed data comes from the numpy oracle."""
import json
import os

import numpy as np
import pytest
import torch


def _tiny_cfg(tie=False):
    from transformers import LlamaConfig
    return LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=32, vocab_size=320, max_position_embeddings=128,
                       tie_word_embeddings=tie)


def _oracle_linear4bit(lin, seed):
    """nf4.Linear4bit on the CPU whose bytes come from the oracle quantiser (first level) + the host-side nested
    statistics quantiser of unsloth_amd.nf4."""
    from oracle.ref_ops import nf4_quantize_np
    from unsloth_amd import nf4
    W = lin.weight.detach().float().numpy()
    packed, absmax = nf4_quantize_np(W, 64)
    absmax = torch.from_numpy(absmax)
    offset = absmax.mean()
    code2 = nf4.create_dynamic_map()
    q, absmax2 = nf4._quantize_blockwise_8bit(absmax - offset, code2, 256)
    state2 = nf4.QuantState(absmax=absmax2, code=code2, blocksize=256, dtype=torch.float32, quant_type=None)
    qs = nf4.QuantState(absmax=q, shape=W.shape, dtype=torch.bfloat16, blocksize=64,
                        code=torch.tensor(nf4.NF4_CODE, dtype=torch.float32), quant_type="nf4", offset=offset,
                        state2=state2)
    return nf4.Linear4bit(lin.in_features, lin.out_features, torch.from_numpy(packed).view(-1, 1), qs, None)


def _quantized_tiny(tie=False):
    from transformers import AutoModelForCausalLM
    from unsloth_amd.models.llama import FastLlamaModel  # noqa: F401  (module import only)
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(_tiny_cfg(tie)).to(torch.bfloat16)
    for li, layer in enumerate(model.model.layers):
        for parent in (layer.self_attn, layer.mlp):
            for n in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"):
                lin = getattr(parent, n, None)
                if isinstance(lin, torch.nn.Linear):
                    setattr(parent, n, _oracle_linear4bit(lin, li))
    return model


@pytest.mark.parametrize("tie", [False, True])
@pytest.mark.parametrize("shard", [False, True])
def test_bnb4bit_round_trip_is_byte_exact(tmp_path, tie, shard):
    from safetensors import safe_open
    from transformers import AutoConfig, AutoModelForCausalLM
    from unsloth_amd import checkpoint as ck
    from unsloth_amd import nf4
    model = _quantized_tiny(tie)
    files = ck.save_pretrained_4bit(model, str(tmp_path), max_shard_size=(60_000 if shard else 1 << 40))
    assert (len(files) > 1) == shard
    # --- key names and dtypes as bitsandbytes / transformers expect them
    keys = {}
    for fn in ck.checkpoint_files(str(tmp_path)):
        with safe_open(os.path.join(str(tmp_path), fn), framework="pt") as f:
            for k in f.keys():
                keys[k] = f.get_tensor(k)
    base = "model.layers.1.mlp.down_proj.weight"
    want = {base, base + ".absmax", base + ".quant_map", base + ".nested_absmax", base + ".nested_quant_map",
            base + ".quant_state.bitsandbytes__nf4"}
    assert want <= set(keys)
    assert keys[base].dtype == torch.uint8 and tuple(keys[base].shape) == (128 * 256 // 2, 1)
    assert keys[base + ".absmax"].dtype == torch.uint8 and keys[base + ".absmax"].numel() == 128 * 256 // 64
    assert keys[base + ".quant_map"].dtype == torch.float32 and keys[base + ".quant_map"].numel() == 16
    assert keys[base + ".nested_quant_map"].numel() == 256
    meta = json.loads(bytes(keys[base + ".quant_state.bitsandbytes__nf4"].tolist()).decode())
    assert meta["quant_type"] == "nf4" and meta["blocksize"] == 64 and meta["nested_blocksize"] == 256
    assert meta["dtype"] == "bfloat16" and tuple(meta["shape"]) == (128, 256) and "nested_offset" in meta
    assert ("lm_head.weight" in keys) == (not tie)
    cfg = AutoConfig.from_pretrained(str(tmp_path))
    assert ck.is_prequantized(cfg)
    qc = cfg.quantization_config if isinstance(cfg.quantization_config, dict) else cfg.quantization_config.to_dict()
    assert qc["bnb_4bit_quant_type"] == "nf4" and qc["bnb_4bit_use_double_quant"] and qc["load_in_4bit"]
    # --- read it back into a storage-less module tree
    import copy
    cfg16 = copy.deepcopy(cfg)
    del cfg16.quantization_config
    with torch.device("meta"):
        fresh = AutoModelForCausalLM.from_config(cfg16, dtype=torch.bfloat16)
    fresh.to_empty(device="cpu")
    missing, unexpected = ck.load_prequantized_(fresh, str(tmp_path), "cpu", torch.bfloat16)
    assert not unexpected and not missing
    a, b = ck.state_dict_4bit(model), ck.state_dict_4bit(fresh)
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k].cpu(), b[k].cpu()), k
    lin = fresh.model.layers[0].self_attn.k_proj
    assert isinstance(lin, nf4.Linear4bit) and lin.weight.quant_state.nested
    assert float(lin.weight.quant_state.offset) == float(model.model.layers[0].self_attn.k_proj.weight.quant_state.offset)
    if tie:
        assert fresh.lm_head.weight is fresh.model.embed_tokens.weight


def test_dequantised_checkpoint_matches_oracle_values(tmp_path):
    """the statistics that come back from disk decode (numpy oracle) to the same matrix as before the trip."""
    from oracle.ref_ops import nf4_dequantize_state
    from unsloth_amd import checkpoint as ck
    model = _quantized_tiny()
    ck.save_pretrained_4bit(model, str(tmp_path))
    side, packed = {}, None
    for name, t in ck.iter_checkpoint_tensors(str(tmp_path)):
        mod, suf = ck._split_quant_key(name)
        if mod == "model.layers.0.mlp.up_proj":
            side[suf] = t
        elif name == "model.layers.0.mlp.up_proj.weight":
            packed = t
    from unsloth_amd.nf4 import QuantState
    qs = QuantState.from_dict(side, "cpu")
    w0 = model.model.layers[0].mlp.up_proj.weight
    want = nf4_dequantize_state(w0.data, w0.quant_state)
    got = nf4_dequantize_state(packed, qs)
    assert torch.equal(got, want)
    assert np.isfinite(got.float().numpy()).all()


def test_load_reconciles_stamped_dtype_detects_missing_tensors_and_rebuilds_rotary(tmp_path):
    """(a) a checkpoint stamped float16 loaded for bf16 compute must end with quant_state.dtype == bf16
    (the GEMM reads the decode scratch as the activation dtype); (b) a tensor the checkpoint lacks is reported as
    missing even though nothing is on the meta device after to_empty(); (c) non-persistent rotary buffers are
    recomputed from the config instead of staying uninitialised."""
    import copy
    from safetensors.torch import load_file, save_file
    from transformers import AutoConfig, AutoModelForCausalLM
    from unsloth_amd import checkpoint as ck
    from unsloth_amd import nf4
    model = _quantized_tiny()
    for m in model.modules():
        if isinstance(m, nf4.Linear4bit):
            m.weight.quant_state.dtype = torch.float16          # how many *-bnb-4bit repos are stamped
    ck.save_pretrained_4bit(model, str(tmp_path))
    cfg = AutoConfig.from_pretrained(str(tmp_path))
    cfg16 = copy.deepcopy(cfg)
    del cfg16.quantization_config

    def fresh():
        with torch.device("meta"):
            m = AutoModelForCausalLM.from_config(cfg16, dtype=torch.bfloat16)
        m.to_empty(device="cpu")
        return m
    f1 = fresh()
    rot = [m for m in f1.modules() if "inv_freq" in getattr(m, "_buffers", {})]
    for m in rot:
        m._buffers["inv_freq"].fill_(float("nan"))             # what uninitialised memory may well hold
    missing, unexpected = ck.load_prequantized_(f1, str(tmp_path), "cpu", torch.bfloat16)
    assert not missing and not unexpected
    qs = f1.model.layers[0].mlp.gate_proj.weight.quant_state
    assert qs.dtype == torch.bfloat16
    want = AutoModelForCausalLM.from_config(cfg16)
    for m, w in zip(rot, [m for m in want.modules() if "inv_freq" in getattr(m, "_buffers", {})]):
        assert torch.equal(m.inv_freq, w.inv_freq)
    # drop one dense tensor from the file: it must come back as missing
    fn = os.path.join(str(tmp_path), ck.checkpoint_files(str(tmp_path))[0])
    sd = load_file(fn)
    victim = "model.layers.1.input_layernorm.weight"
    assert victim in sd
    del sd[victim]
    save_file(sd, fn, metadata={"format": "pt"})
    missing, _ = ck.load_prequantized_(fresh(), str(tmp_path), "cpu", torch.bfloat16)
    assert missing == [victim]


def test_nf4_code_table_is_the_published_quantile_construction():
    """The 16 NF4 levels are not free constants: bitsandbytes builds them (functional.create_normal_map, offset 0.9677083,
    the QLoRA paper's "k-bit NormalFloat") from quantiles of N(0, 1) -- 8 positive levels norm.ppf(linspace(offset, 0.5, 9)[:-1]),
    7 negative ones -norm.ppf(linspace(offset, 0.5, 8)[:-1]), an exact zero, normalised by the largest. Re-deriving them here
    (torch.linspace in fp32, as bitsandbytes does) must give, BIT FOR BIT, the table of the oracle (oracle/ref_ops.py), of the
    host module (unsloth_amd/nf4.py) and of the HIP kernels (csrc/nf4.hip kNF4): the one part of the third-party NF4 format
    that can be pinned to its published algorithm without bitsandbytes installed."""
    import re
    import numpy as np
    from scipy.stats import norm
    from oracle.ref_ops import NF4_CODE as ORACLE_CODE
    from unsloth_amd.nf4 import NF4_CODE as HOST_CODE
    offset = 0.9677083
    pos = norm.ppf(torch.linspace(offset, 0.5, 9)[:-1]).tolist()
    neg = (-norm.ppf(torch.linspace(offset, 0.5, 8)[:-1])).tolist()
    levels = torch.tensor(sorted(pos + [0.0] + neg))
    levels = (levels / levels.max()).numpy().astype(np.float32)
    assert levels.shape == (16,) and levels[7] == 0.0 and levels[0] == -1.0 and levels[15] == 1.0
    assert np.array_equal(levels, np.asarray(ORACLE_CODE, dtype=np.float32))
    assert np.array_equal(levels, np.asarray(HOST_CODE, dtype=np.float32))
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(ROOT, "unsloth_amd", "csrc", "nf4.hip")).read()
    body = re.search(r"kNF4\[16\]\s*=\s*\{([^}]*)\}", src).group(1)
    hip = np.array([float(t.strip().rstrip("f")) for t in body.split(",") if t.strip()], dtype=np.float32)
    assert np.array_equal(levels, hip)
    # the decode GEMV carries its own copy of the table (csrc/decode.hip)
    dec = open(os.path.join(ROOT, "unsloth_amd", "csrc", "decode.hip")).read()
    m = re.search(r"kNF4d\[16\]\s*=\s*\{([^}]*)\}", dec)
    assert m is not None
    if m:
        dtab = np.array([float(t.strip().rstrip("f")) for t in m.group(1).split(",") if t.strip()], dtype=np.float32)
        assert np.array_equal(levels, dtab)
