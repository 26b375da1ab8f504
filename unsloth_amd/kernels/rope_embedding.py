"""RoPE through the HIP kernel; mirror of unsloth/kernels/rope_embedding.py.

  Fast_RoPE_Embedding     (:169-261)  dense [B,T,H,D] rows, position = row % seqlen, in place
  Fast_RoPE_Embedding_QK  (:283-399)  Q,K as [B,H,T,D] (possibly strided views), optional int32
                                      per-token gather indices, one launch for Q and K
  fast_rope_embedding     (:265-280)  dispatch on rope_embedding_indices
  fast_mrope_embedding                Qwen2-VL multimodal RoPE (3 position streams) on the same kernel; oracle =
                                      transformers' apply_multimodal_rotary_pos_emb
  inplace_rope_embedding  (:435-439)  explicit positions -> the same kernel with gather indices (the reference's
                                      torch formulation Slow_RoPE_Embedding :402-432 has no counterpart here: the
                                      oracle restates it, oracle/ref_ops.py)

MI355X difference: the kernel takes element strides, so `fast_rope_embedding` without indices
does NOT pay the reference's `Q.transpose(1,2).contiguous()` copies (:276-277): it rotates the
[B,H,T,D] views in place through the strided entry point. Backward = same kernel with sin -> -sin.
"""
import torch

from .. import _lib


def _tables(cos, sin):
    cos, sin = cos.squeeze(), sin.squeeze()
    if cos.dim() != 2 or cos.stride(1) != 1 or sin.stride(1) != 1:
        cos, sin = cos.reshape(-1, cos.shape[-1]).contiguous(), sin.reshape(-1, sin.shape[-1]).contiguous()
    return cos, sin


def _launch_qk(Q, K, cos, sin, idx, backward):
    batch, n_heads_Q, seq_len, head_dim = Q.shape
    n_heads_K = K.shape[1] if K is not None else 0
    if Q.stride(3) != 1 or (K is not None and K.stride(3) != 1):
        raise ValueError("head_dim must be the contiguous dimension")
    with _lib.device_ctx(Q):
        rc = _lib.lib().uamd_rope_embedding_qk(
            _lib.ptr(Q), Q.stride(0), Q.stride(1), Q.stride(2),
            _lib.ptr(K), *( (K.stride(0), K.stride(1), K.stride(2)) if K is not None else (0, 0, 0) ),
            _lib.ptr(cos), cos.stride(0), _lib.ptr(sin), sin.stride(0), _lib.ptr(idx),
            batch, seq_len, n_heads_Q, n_heads_K, head_dim, int(backward),
            _lib.dtype_code(Q.dtype), _lib.dtype_code(cos.dtype), _lib.stream_of(Q))
    _lib.check(rc, "uamd_rope_embedding_qk")


class Fast_RoPE_Embedding(torch.autograd.Function):
    """Q: [batch, seq_len, n_heads, head_dim] contiguous; rotated in place (:169-261)."""

    @staticmethod
    def forward(ctx, Q, cos, sin):
        _lib.require_gpu(Q, cos, sin)
        cos, sin = _tables(cos, sin)
        batch, seq_len, n_heads, head_dim = Q.shape
        assert seq_len <= cos.shape[0]
        Q = Q.reshape(batch * seq_len, n_heads * head_dim)
        if Q.stride(1) != 1:
            Q = Q.contiguous()
        Fast_RoPE_Embedding._run(Q, cos, sin, seq_len, n_heads, head_dim, False)
        ctx.cos, ctx.sin = cos, sin
        return Q.reshape(batch, seq_len, n_heads, head_dim)

    @staticmethod
    def _run(Q2d, cos, sin, seq_len, n_heads, head_dim, backward):
        with _lib.device_ctx(Q2d):
            rc = _lib.lib().uamd_rope_embedding(
                _lib.ptr(Q2d), Q2d.stride(0), _lib.ptr(cos), cos.stride(0), _lib.ptr(sin), sin.stride(0),
                Q2d.shape[0], seq_len, n_heads, head_dim, int(backward), _lib.dtype_code(Q2d.dtype),
                _lib.dtype_code(cos.dtype), _lib.stream_of(Q2d))
        _lib.check(rc, "uamd_rope_embedding")

    @staticmethod
    def backward(ctx, dY):
        batch, seq_len, n_heads, head_dim = dY.shape
        dY = dY.reshape(batch * seq_len, n_heads * head_dim)
        if dY.stride(1) != 1:
            dY = dY.contiguous()
        Fast_RoPE_Embedding._run(dY, ctx.cos, ctx.sin, seq_len, n_heads, head_dim, True)
        return dY.reshape(batch, seq_len, n_heads, head_dim), None, None


class Fast_RoPE_Embedding_QK(torch.autograd.Function):
    """Q [B,Hq,T,D], K [B,Hk,T,D]; rope_indices int32 [B*T] or None (:283-399). In place; strided
    views (e.g. the transposed halves of a fused QKV GEMM output) are rotated where they live."""

    @staticmethod
    def forward(ctx, Q, K, cos, sin, rope_indices):
        _lib.require_gpu(Q, K, cos, sin)
        has_indices = rope_indices is not None
        cos, sin = _tables(cos, sin)
        Q_out = Q if Q.stride(-1) == 1 else Q.contiguous()
        K_out = K if K.stride(-1) == 1 else K.contiguous()
        idx = None
        if has_indices:
            # rope_embedding.py:295-297: int32 on the device
            idx = rope_indices.reshape(-1).to(dtype=torch.int32, device=Q.device).contiguous()
            assert idx.numel() == Q.shape[0] * Q.shape[2]
        _launch_qk(Q_out, K_out, cos, sin, idx, False)
        ctx.cos, ctx.sin, ctx.idx = cos, sin, idx
        # Like the reference (:290-294 "Inplace rotary embedding is generally fine") the rotation is
        # written through the raw pointer and the input object is returned; nothing upstream needs
        # the un-rotated projections (LoRA_QKV saves X, not Q/K).
        return Q_out, K_out

    @staticmethod
    def backward(ctx, dQ, dK):
        dQ_out = dQ if dQ.stride(-1) == 1 else dQ.contiguous()
        dK_out = dK if dK.stride(-1) == 1 else dK.contiguous()
        _launch_qk(dQ_out, dK_out, ctx.cos, ctx.sin, ctx.idx, True)
        return dQ_out, dK_out, None, None, None


@torch.compiler.disable
def fast_rope_embedding(Q, K, cos, sin, rope_embedding_indices=None):
    """rope_embedding.py:265-280. Q [B,Hq,T,D], K [B,Hk,T,D] -> (Q, K) rotated."""
    return Fast_RoPE_Embedding_QK.apply(Q, K, cos, sin, rope_embedding_indices)


class Fast_MRoPE_Embedding_QK(torch.autograd.Function):
    """Multimodal RoPE (Qwen2-VL text tower): Q [B,Hq,T,D], K [B,Hk,T,D] rotated in place with three
    position streams. `positions3` int [3, B, T] (temporal, height, width), `mrope_section` = (s_t, s_h, s_w) rotary
    pairs per stream (Qwen2-VL-7B: 16, 24, 24). cos/sin: the ordinary [max_position, D] table of the text model.
    Semantics and rounding points = transformers' apply_multimodal_rotary_pos_emb (which the tests use as oracle)."""

    @staticmethod
    def forward(ctx, Q, K, cos, sin, positions3, mrope_section):
        _lib.require_gpu(Q, K, cos, sin)
        cos, sin = _tables(cos, sin)
        s_t, s_h, s_w = (int(x) for x in mrope_section)
        assert s_t + s_h + s_w == Q.shape[-1] // 2, "mrope_section must cover head_dim / 2 rotary pairs"
        Q_out = Q if Q.stride(-1) == 1 else Q.contiguous()
        K_out = K if K.stride(-1) == 1 else K.contiguous()
        B, _, T, _ = Q.shape
        pos3 = positions3.to(device=Q.device, dtype=torch.int32).reshape(3, -1).contiguous()
        assert pos3.shape[1] == B * T
        ctx.args = (cos, sin, pos3, s_t, s_h)
        Fast_MRoPE_Embedding_QK._run(Q_out, K_out, cos, sin, pos3, s_t, s_h, False)
        return Q_out, K_out

    @staticmethod
    def _run(Q, K, cos, sin, pos3, s_t, s_h, backward):
        batch, n_heads_Q, seq_len, head_dim = Q.shape
        with _lib.device_ctx(Q):
            rc = _lib.lib().uamd_rope_embedding_qk_mrope(
                _lib.ptr(Q), Q.stride(0), Q.stride(1), Q.stride(2), _lib.ptr(K), K.stride(0), K.stride(1), K.stride(2),
                _lib.ptr(cos), cos.stride(0), _lib.ptr(sin), sin.stride(0), _lib.ptr(pos3), s_t, s_h, batch, seq_len,
                n_heads_Q, K.shape[1], head_dim, int(backward), _lib.dtype_code(Q.dtype), _lib.dtype_code(cos.dtype),
                _lib.stream_of(Q))
        _lib.check(rc, "uamd_rope_embedding_qk_mrope")

    @staticmethod
    def backward(ctx, dQ, dK):
        cos, sin, pos3, s_t, s_h = ctx.args
        dQ = dQ if dQ.stride(-1) == 1 else dQ.contiguous()
        dK = dK if dK.stride(-1) == 1 else dK.contiguous()
        Fast_MRoPE_Embedding_QK._run(dQ, dK, cos, sin, pos3, s_t, s_h, True)
        return dQ, dK, None, None, None, None


@torch.compiler.disable
def fast_mrope_embedding(Q, K, cos, sin, positions3, mrope_section):
    """(Q, K) rotated by multimodal RoPE; see Fast_MRoPE_Embedding_QK."""
    return Fast_MRoPE_Embedding_QK.apply(Q, K, cos, sin, positions3, tuple(mrope_section))


def inplace_rope_embedding(Q, K, cos, sin, position_ids):
    """rope_embedding.py:435-439 (the reference's decode-time entry: rotate Q and K [B,H,T,D] in place at explicit
    `position_ids`). Here it is the SAME HIP kernel as training, with the positions as its per-token gather
    indices; there is no torch-op formulation of RoPE anywhere on the product path."""
    if position_ids is None:
        return fast_rope_embedding(Q, K, cos, sin, None)
    bsz, q_len = Q.shape[0], Q.shape[2]
    idx = position_ids.to(device=Q.device, dtype=torch.int32)
    if idx.dim() == 1:
        idx = idx.unsqueeze(0)
    idx = idx.expand(bsz, q_len).reshape(-1)
    return fast_rope_embedding(Q, K, cos, sin, idx)
