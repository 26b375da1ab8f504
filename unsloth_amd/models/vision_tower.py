"""Qwen2-VL's vision tower on the hand kernels (BASELINE config 4).

transformers' Qwen2VLVisionBlock = LayerNorm -> VisionAttention (qkv Linear, 2-D RoPE, NON-CAUSAL attention inside every image /
frame's `cu_seqlens` window, proj Linear) -> LayerNorm -> VisionMlp (fc1, QuickGELU, fc2). The reference hands that module tree to
the unsloth_zoo compiler (unsloth/models/vision.py:881-1990; third party). Here the blocks keep HF's modules and parameters, and
their forwards are replaced per instance (the way the language tower's attention / MLP forwards are, models/llama.py):

  * attention: q | k | v stay where the qkv GEMM wrote them ([S, 3, H, D] -> three strided [1, S, H, D] views), the rotary
    embedding (fp32 tables of the 2-D patch positions, row = patch) rotates q and k IN PLACE through csrc/rope_embedding.hip,
    and csrc/attention.hip runs the bidirectional attention inside the `cu_seqlens` windows (kernels/attention.document_band;
    head_dim 80 is native to the kernels since round 6: no padded copies of q, k, v, o, dO) -- in round 3 this was torch SDPA:
    aotriton's flash kernels at 14 % (forward) and 6 % (backward) of the MFMA peak, 20 % of the config-4 step
    (profiles/r04m_config4_kernel_stats.csv: bwd_kernel_dk_dv 1.07 ms, bwd_kernel_dq 0.38 ms, attn_fwd 0.24 ms per block);
  * patch embedding: the Conv3d with kernel == stride as ONE GEMM over the flattened patches (no MIOpen convolution);
  * MLP: fc1 / fc2 are the module's own linears (LoRA_W on the MFMA GEMM once adapters are attached), QuickGELU is one streaming
    HIP kernel each way (kernels/quick_gelu.py) instead of three torch elementwise kernels forward and more backward.
CPU tensors / fp32 activations fall through to transformers' forward (the CPU tests of tests/test_vision.py)."""
from types import MethodType

import torch

from ..kernels import attention as _flash
from ..kernels.quick_gelu import fast_quick_gelu
from ..kernels.rope_embedding import fast_rope_embedding

_BAND_CACHE = {}


def _band_of(cu_seqlens, S, device):
    """(lo, hi) of the `cu_seqlens` windows, built once per forward (every block of the tower passes the same tensor)."""
    key = (cu_seqlens.data_ptr(), int(cu_seqlens.numel()), S, device)
    hit = _BAND_CACHE.get("last")
    if hit is not None and hit[0] == key and hit[1] is cu_seqlens:
        return hit[2]
    lengths = (cu_seqlens[1:] - cu_seqlens[:-1]).to(torch.int64)
    band = None if lengths.numel() == 1 else _flash.document_band(S, batch=1, seq_lengths=lengths, device=device)
    _BAND_CACHE["last"] = (key, cu_seqlens, band)
    return band


def vision_attention_fast_forward(self, hidden_states, cu_seqlens, position_embeddings=None, max_seqlen=None, **kwargs):
    """Qwen2-VL VisionAttention.forward on the HIP kernels (see the module docstring)."""
    if (not hidden_states.is_cuda) or hidden_states.dtype not in (torch.bfloat16, torch.float16) or position_embeddings is None \
            or self.head_dim > 128 or (self.head_dim & 1):
        return self._uamd_hf_forward(hidden_states, cu_seqlens, position_embeddings=position_embeddings, max_seqlen=max_seqlen,
                                     **kwargs)
    S = hidden_states.shape[0]
    H, D = self.num_heads, self.head_dim
    qkv = self.qkv(hidden_states).view(1, S, 3, H, D)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                         # [1, S, H, D] views of the GEMM output
    cos, sin = position_embeddings                                             # [S, D] fp32: row = patch, first D/2 columns used
    qr, kr = fast_rope_embedding(q.transpose(1, 2), k.transpose(1, 2), cos.float(), sin.float(), None)
    o = _flash.flash_attention(qr.transpose(1, 2), kr.transpose(1, 2), v, float(self.scaling), _band_of(cu_seqlens, S, qkv.device),
                               False)
    return self.proj(o.reshape(S, H * D))


def vision_mlp_fast_forward(self, x):
    if (not x.is_cuda) or x.dtype not in (torch.bfloat16, torch.float16) or not getattr(self, "_uamd_quick_gelu", False):
        return self._uamd_hf_forward(x)
    return self.fc2(fast_quick_gelu(self.fc1(x)))


def patch_embed_fast_forward(self, hidden_states):
    """PatchEmbed.forward: a Conv3d whose kernel equals its stride IS a linear map of the flattened patch -- [n, C * t * p * p] @
    W.view(embed_dim, -1)^T on the MFMA GEMM (kernels/fast_dense.Dense_W) instead of a MIOpen convolution (whose first call on a
    fresh box spends minutes in kernel search / compilation)."""
    W = self.proj.weight
    if (not hidden_states.is_cuda) or W.dtype not in (torch.bfloat16, torch.float16) or self.proj.bias is not None \
            or (W[0].numel() % 8):
        return self._uamd_hf_forward(hidden_states)
    from ..kernels.fast_dense import Dense_W
    x = hidden_states.reshape(-1, W[0].numel()).to(W.dtype)
    return Dense_W.apply(x, W.view(W.shape[0], -1), None).view(-1, self.embed_dim)


def merger_fast_forward(self, x):
    """PatchMerger.forward (LayerNorm -> Linear -> GELU -> Linear over 2 x 2 merged patches): the two frozen linears on the MFMA
    GEMM (kernels/fast_dense.Dense_W: Y = X W^T + b forward, dX = dY W backward) instead of F.linear -- these were the last
    hipBLASLt launches of the config-4 step (profiles/r06zh_config4_kernel_stats.csv: three per step)."""
    l0, l1 = self.mlp[0], self.mlp[2]
    if (not x.is_cuda) or x.dtype not in (torch.bfloat16, torch.float16) or l0.weight.dtype != x.dtype or l1.weight.dtype != x.dtype:
        return self._uamd_hf_forward(x)
    from ..kernels.fast_dense import Dense_W
    h = self.ln_q(x).view(-1, self.hidden_size)
    h = self.mlp[1](Dense_W.apply(h, l0.weight, l0.bias))
    return Dense_W.apply(h, l1.weight, l1.bias)


def patch_vision_tower(visual):
    """Install the fast forwards on every block of a Qwen2VisionTransformerPretrainedModel. Returns the number of blocks patched."""
    n = 0
    pe = getattr(visual, "patch_embed", None)
    if pe is not None and isinstance(getattr(pe, "proj", None), torch.nn.Conv3d) and not hasattr(pe, "_uamd_hf_forward") \
            and tuple(pe.proj.kernel_size) == tuple(pe.proj.stride) and tuple(pe.proj.padding) == (0, 0, 0):
        pe._uamd_hf_forward = pe.forward
        pe.forward = MethodType(patch_embed_fast_forward, pe)
    mg = getattr(visual, "merger", None)
    seq = getattr(mg, "mlp", None)
    if (mg is not None and isinstance(seq, torch.nn.Sequential) and len(seq) == 3 and type(seq[0]) is torch.nn.Linear
            and type(seq[2]) is torch.nn.Linear and hasattr(mg, "ln_q") and hasattr(mg, "hidden_size")
            and seq[0].in_features % 8 == 0 and seq[0].out_features % 8 == 0 and seq[2].out_features % 8 == 0
            and not hasattr(mg, "_uamd_hf_forward")):
        mg._uamd_hf_forward = mg.forward
        mg.forward = MethodType(merger_fast_forward, mg)
    for blk in getattr(visual, "blocks", []):
        attn, mlp = getattr(blk, "attn", None), getattr(blk, "mlp", None)
        if attn is not None and hasattr(attn, "qkv") and hasattr(attn, "proj") and not hasattr(attn, "_uamd_hf_forward"):
            attn._uamd_hf_forward = attn.forward
            attn.forward = MethodType(vision_attention_fast_forward, attn)
            n += 1
        if mlp is not None and hasattr(mlp, "fc1") and hasattr(mlp, "fc2") and not hasattr(mlp, "_uamd_hf_forward"):
            mlp._uamd_quick_gelu = type(getattr(mlp, "act", None)).__name__ == "QuickGELUActivation"
            mlp._uamd_hf_forward = mlp.forward
            mlp.forward = MethodType(vision_mlp_fast_forward, mlp)
    return n
