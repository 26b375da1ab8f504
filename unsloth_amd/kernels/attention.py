"""Causal GQA flash attention through the HIP kernels (csrc/attention.hip).

The reference delegates this step to flash-attn / xformers / torch SDPA (`run_attention`,
unsloth/utils/attention_dispatch.py:298-617, called from unsloth/models/llama.py:757); SURVEY 8(f1). Here it is a
pair of hand-written CDNA4 kernels that read Q/K/V in the [B, T, H, D] layout the QKV GEMM produced (any strides,
d contiguous) and write O as [B, T, Hq, D] == the o_proj input, so none of the reference's transposes
(llama.py:276-277) or `.contiguous()` copies exist.
"""
import ctypes
import math

import torch

from .. import _lib

_I64x12 = ctypes.c_int64 * 12
_I64x24 = ctypes.c_int64 * 24


_GROUPS = (1, 2, 3, 4, 5, 6, 7, 8)   # query heads per KV head the kernels take (csrc/attention.hip: the forward and dQ kernels run a
                                     # KV head's heads in groups of 8, 4, 2 or 1 over virtual KV heads, the dK / dV kernel in passes of
                                     # 4, 2 and 1 heads -- rounds 2-5 zero-padded 3 / 5 / 6 / 7 to 4 / 8 through copies of Q, O and dO)


def native(q, k, v):
    """Shapes the kernels take as they are: G = Hq / Hk in 1 .. 8 and a head_dim that is a multiple of 8 up to 128. Below 128
    (TinyLlama / Llama-3.2-1B: 64, Qwen2-VL's vision tower: 80) the kernels still work on 256-byte rows but read nothing past a
    head's D elements as data (csrc/attention.hip AttnArgs::D): no padded copies of q, k, v, o, do."""
    return (q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and 8 <= q.shape[-1] <= 128 and q.shape[-1] % 8 == 0
            and k.shape[2] > 0 and q.shape[2] % k.shape[2] == 0 and (q.shape[2] // k.shape[2]) in _GROUPS)


def supported(q, k, v):
    """Native shapes, plus the ones that run on the same kernels after zero-padding (`_pad_qkv`): head dims below 128 that are
    not a multiple of 8."""
    return (q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and 0 < q.shape[-1] <= 128
            and k.shape[2] > 0 and q.shape[2] % k.shape[2] == 0 and (q.shape[2] // k.shape[2]) <= 8)


_ZEROS = {}         # (dtype, device) -> ONE flat zero buffer that is only ever READ (the padding operand of the concatenations below)


def _zeros(shape, dtype, device):
    """A zero tensor of `shape` as a view of a per-(dtype, device) flat buffer that only ever GROWS (a buffer is never freed
    while a concatenation on some stream may still read it, and variable shapes -- ViT grids, packed rows -- share it instead of
    pinning one tensor per shape)."""
    n = 1
    for d in shape:
        n *= int(d)
    key = (dtype, torch.device(device))
    z = _ZEROS.get(key)
    if z is None or z.numel() < n:
        grown = torch.zeros(max(n, 2 * z.numel() if z is not None else 0), dtype=dtype, device=device)
        z = _ZEROS[key] = grown          # the old buffer stays alive as long as a view of it does
    return z[:n].view(tuple(int(d) for d in shape))


def _pad_heads(x, G, Gp, D):
    """[B,T,Hk*G,D] (any strides, d contiguous) -> contiguous [B,T,Hk*Gp,128], zeros in the added head slots / columns. ONE
    concatenation with a cached zero operand per padded axis (torch.nn.functional.pad is a fill + a copy: two launches and a
    write of the whole result where one launch writes it once -- 950 of config 4's launches per step were these, round 4)."""
    B, T, Hq, _ = x.shape
    Hk = Hq // G
    if D != 128:
        x = torch.cat([x, _zeros((B, T, Hq, 128 - D), x.dtype, x.device)], dim=-1)
    if Gp != G:
        x = torch.cat([x.reshape(B, T, Hk, G, 128), _zeros((B, T, Hk, Gp - G, 128), x.dtype, x.device)], dim=3)
    return x.reshape(B, T, Hk * Gp, 128)


def _pad_qkv(q, k, v):
    """Zero-padding onto a native shape. Head dim D < 128: extra zero columns change neither Q K^T nor the real columns
    of P V. (Every group size 1 .. 8 is native since round 6, so Gp == G; the head padding below is what a group size outside
    `_GROUPS` would take.) Group size G not in _GROUPS: every KV group gets Gp - G extra query heads that are all zero -- their
    scores are 0, their dO is 0 (the caller never sees their output), so they add exactly nothing to dK / dV
    (dV += P^T dO = 0; dP = dO V^T = 0 and Delta = 0 give dS = 0). Returns (qp [B,T,Hk*Gp,128], kp, vp, G, Gp)."""
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    G = Hq // Hk
    Gp = next(g for g in _GROUPS if g >= G)
    qp = _pad_heads(q, G, Gp, D)
    kp = _pad_heads(k, 1, 1, D) if D != 128 else k
    vp = _pad_heads(v, 1, 1, D) if D != 128 else v
    return qp, kp, vp, G, Gp


def _pad_like_q(x, G, Gp):
    """[B,T,Hq,D] -> [B,T,Hk*Gp,128] with the same zero padding as the queries (for O and dO)."""
    return _pad_heads(x, G, Gp, x.shape[-1])


def _strides(*ts):
    out = []
    for t in ts:
        assert t.stride(3) == 1, "head_dim must be contiguous"
        out += [t.stride(0), t.stride(1), t.stride(2)]
    return (_I64x12 if len(ts) == 4 else _I64x24)(*out)


def _pad32(T):
    return (T + 31) // 32 * 32


def attention_band(T, batch=1, seq_lengths=None, sliding_window=None, device="cpu"):
    """Band of the block-diagonal causal mask as two int32 [batch, T] arrays (lo, hi): query q attends keys
    lo[q] <= key <= q, i.e. key is seen by queries key <= q <= hi[key] (positions inside the batch row).
    `seq_lengths`: lengths of the documents packed back to back over the FLATTENED batch of batch*T tokens (the
    reference's `packed_seq_lengths`, utils/packing.py:586-606: its varlen attention runs on the flattened rows
    with cu_seqlens = [0, cumsum(lengths)]); tokens past sum(lengths) form one more document. A batch row is its own
    attention problem here, so a document that straddles a row boundary is cut there (packing collators end every
    document inside its row). `sliding_window` W: additionally q - key < W (packing.py:679-683,
    attention_dispatch.py:292). Integer work, done with torch ops on `device`; exact."""
    total = batch * T
    g = torch.arange(total, dtype=torch.int64, device=device)       # flat token index
    row0 = (g // T) * T                                             # first flat index of the token's batch row
    if seq_lengths is not None:
        lens = torch.as_tensor(seq_lengths, dtype=torch.int64, device=device).flatten()
        lens = lens[lens > 0]
        ends = torch.cumsum(lens, 0).clamp_(max=total)             # exclusive end of every document
        doc = torch.searchsorted(ends, g, right=True)              # document of every token
        ends = torch.cat([ends, ends.new_full((1,), total)])       # trailing (padding) tokens: one more document
        starts = torch.cat([ends.new_zeros(1), ends[:-1]])
        lo = torch.maximum(starts[doc], row0) - row0
        hi = torch.minimum(ends[doc] - 1, row0 + (T - 1)) - row0
    else:
        lo, hi = torch.zeros_like(g), torch.full_like(g, T - 1)
    if sliding_window is not None and sliding_window > 0:
        pos = g - row0
        lo = torch.maximum(lo, pos - (sliding_window - 1))
        hi = torch.minimum(hi, pos + (sliding_window - 1))
    return lo.to(torch.int32).view(batch, T).contiguous(), hi.to(torch.int32).view(batch, T).contiguous()


def document_band(T, batch=1, seq_lengths=None, device="cpu"):
    """(lo, hi) int32 [batch, T] of NON-CAUSAL attention inside documents: position t attends every position of its own document,
    lo[t] = first, hi[t] = last position of it. `seq_lengths`: document lengths back to back over the flattened batch (the
    `cu_seqlens` windows of a vision tower: one entry per image / frame); None = every batch row is one document. Documents are
    cut at row boundaries like attention_band's. Integer work with torch ops; exact."""
    total = batch * T
    g = torch.arange(total, dtype=torch.int64, device=device)
    row0 = (g // T) * T
    if seq_lengths is not None:
        lens = torch.as_tensor(seq_lengths, dtype=torch.int64, device=device).flatten()
        lens = lens[lens > 0]
        ends = torch.cumsum(lens, 0).clamp_(max=total)
        doc = torch.searchsorted(ends, g, right=True)
        ends = torch.cat([ends, ends.new_full((1,), total)])
        starts = torch.cat([ends.new_zeros(1), ends[:-1]])
        lo = torch.maximum(starts[doc], row0) - row0
        hi = torch.minimum(ends[doc] - 1, row0 + (T - 1)) - row0
    else:
        lo, hi = torch.zeros_like(g), torch.full_like(g, T - 1)
    return lo.to(torch.int32).view(batch, T).contiguous(), hi.to(torch.int32).view(batch, T).contiguous()


def padding_mask_documents(attention_mask):
    """A key-padding mask [B, T] (non-zero = real token) whose real tokens are CONTIGUOUS in every row (right padding,
    left padding, or both) as packed-document lengths over the flattened batch: per row [pad_left, real, pad_right].
    Under a causal mask a real query then attends exactly the real keys before it -- the semantics of the SDPA /
    flash-attn padding paths (attention_dispatch.py:560-617) -- and a padding query (whose output and gradient are
    never used: its label is -100) attends the padding run it sits in, so no row of the softmax is empty (a left-padded
    row is all -inf, i.e. NaN, under a dense key mask). Returns int32 [3 B] lengths (zeros included; attention_band drops
    them) or None when some row has holes. Integer work on the mask's device; one host sync for the contiguity test."""
    m = attention_mask != 0
    B, T = m.shape
    n = m.sum(1)
    first = torch.where(n > 0, m.int().argmax(1), torch.zeros_like(n))
    last = T - 1 - m.flip(1).int().argmax(1)
    if not bool(((last - first + 1 == n) | (n == 0)).all()):
        return None
    return torch.stack([first, n, T - first - n], 1).reshape(-1).to(torch.int32)


def _band_ptrs(band, B, T, dev):
    if band is None:
        return None, None
    lo, hi = band
    for x in (lo, hi):
        assert x.dtype == torch.int32 and x.shape == (B, T) and x.is_contiguous() and x.device == dev, \
            "band = (lo, hi): contiguous int32 [B, T] on the activations' device"
    return _lib.ptr(lo), _lib.ptr(hi)


def attn_forward(q, k, v, scale=None, band=None, causal=True, keep_padded=None):
    """q [B,T,Hq,128], k/v [B,T,Hk,128] (strided views are fine) -> (o [B,T,Hq,128] contiguous, lse [B,Hq,T] fp32).
    `band` = (lo, hi) from attention_band() restricts the causal mask to packed documents / a sliding window.
    causal=False: bidirectional attention inside the documents of `band` = document_band(...) (None: each row one document)."""
    _lib.require_gpu(q, k, v)
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if not native(q, k, v):
        assert supported(q, k, v), "head_dim <= 128 and at most 8 query heads per KV head"
        qp, kp, vp, G, Gp = _pad_qkv(q, k, v)
        op, lsep = _forward_native(qp, kp, vp, scale, band) if causal else _forward_native(qp, kp, vp, scale, band, False)
        if keep_padded is not None:       # the caller saves the padded operands for attn_backward(padded=...) instead of q, k, v, o
            keep_padded.extend([qp, kp, vp, op, lsep])
        o = op.view(B, T, Hk, Gp, 128)[:, :, :, :G, :D].reshape(B, T, Hq, D)
        Tp = _pad32(T)
        lse = torch.as_strided(lsep, (B, Hk, Gp, Tp), (Hk * Gp * Tp, Gp * Tp, Tp, 1))[:, :, :G].reshape(B, Hq, Tp)
        return o, lse[:, :, :T]
    return _forward_native(q, k, v, scale, band) if causal else _forward_native(q, k, v, scale, band, False)


def _forward_native(q, k, v, scale, band, causal=True):
    """The launch itself: head_dim a multiple of 8 up to 128, G in 1 .. 8."""
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    o = torch.empty((B, T, Hq, D), dtype=q.dtype, device=q.device)
    Tp = _pad32(T)
    lse = (torch.empty if Tp == T else torch.zeros)((B, Hq, Tp), dtype=torch.float32, device=q.device)
    if not causal and band is None:
        band = document_band(T, batch=B, device=q.device)
    lo, hi = _band_ptrs(band, B, T, q.device)
    with _lib.device_ctx(q):
        if causal:
            rc = _lib.lib().uamd_attn_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(lse),
                                          _strides(q, k, v, o), B, T, Hq, Hk, D, Tp, float(scale), 1, lo,
                                          _lib.dtype_code(q.dtype), _lib.stream_of(q))
        else:
            rc = _lib.lib().uamd_attn_fwd_band(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(lse),
                                               _strides(q, k, v, o), B, T, Hq, Hk, D, Tp, float(scale), 0, lo, hi,
                                               _lib.dtype_code(q.dtype), _lib.stream_of(q))
    _lib.check(rc, "uamd_attn_fwd")
    return o, lse[:, :, :T]


def attn_backward(do, q, k, v, o, lse, scale=None, band=None, causal=True, padded=None):
    """Gradients of attn_forward: (dq [B,T,Hq,D], dk, dv [B,T,Hk,D]) in q's dtype, column blocks of one buffer. `lse` is the view
    attn_forward returned (its storage is padded to a multiple of 32 positions). Two launches, deterministic.
    `padded` = what attn_forward(keep_padded=[...]) collected for a shape that runs zero-padded: (qp, kp, vp, op, lsep) are used as
    they are (q, k, v then only give the shapes; o and lse may be None) and only dO is padded here."""
    _lib.require_gpu(do, q, k, v)
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    Tp = _pad32(T)
    assert padded is not None or (lse.stride(1) == Tp and lse.stride(2) == 1), "pass the LSE returned by attn_forward"
    if not native(q, k, v):
        extra = () if causal else (False,)
        if padded is not None:
            qp, kp, vp, op, lsep = padded
            G = Hq // Hk
            Gp = qp.shape[2] // Hk
            dqp, dkp, dvp = _backward_native(_pad_like_q(do, G, Gp), qp, kp, vp, op, lsep, scale, band, *extra)
        else:
            qp, kp, vp, G, Gp = _pad_qkv(q, k, v)
            lse_full = torch.as_strided(lse, (B, Hq, Tp), (Hq * Tp, Tp, 1))
            lsep = lse_full if Gp == G else torch.cat([lse_full.view(B, Hk, G, Tp), _zeros((B, Hk, Gp - G, Tp), lse.dtype, lse.device)],
                                                      dim=2).view(B, Hk * Gp, Tp)
            dqp, dkp, dvp = _backward_native(_pad_like_q(do, G, Gp), qp, kp, vp, _pad_like_q(o, G, Gp), lsep[:, :, :T], scale, band,
                                             *extra)
        # same contract as below: dQ | dK | dV as column blocks of ONE buffer
        dqkv = torch.empty((B, T, (Hq + 2 * Hk) * D), dtype=q.dtype, device=q.device)
        dq = dqkv[..., :Hq * D].view(B, T, Hq, D)
        dk = dqkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
        dv = dqkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
        dq.view(B, T, Hk, G, D).copy_(dqp.view(B, T, Hk, Gp, 128)[:, :, :, :G, :D])
        dk.copy_(dkp[..., :D])
        dv.copy_(dvp[..., :D])
        return dq, dk, dv
    return _backward_native(do, q, k, v, o, lse, scale, band) if causal else _backward_native(do, q, k, v, o, lse, scale, band, False)


def _backward_native(do, q, k, v, o, lse, scale, band, causal=True):
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    Tp = _pad32(T)
    if do.stride(3) != 1:
        do = do.contiguous()
    # dQ | dK | dV side by side in ONE [B, T, (Hq + 2 Hk) D] buffer, the layout of a fused QKV projection output:
    # the q/k/v projections' backward can then run dX as a single K-concatenated GEMM (kernels/utils.py)
    dqkv = torch.empty((B, T, (Hq + 2 * Hk) * D), dtype=q.dtype, device=q.device)
    dq = dqkv[..., :Hq * D].view(B, T, Hq, D)
    dk = dqkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    dv = dqkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    # scratch of the two launches: plane 0 = -Delta = -rowsum(dO * O), plane 1 = LSE * log2(e) (written by the dQ kernel,
    # read by the dK/dV kernel's LDS-DMA)
    delta = (torch.empty if Tp == T else torch.zeros)((2, B, Hq, Tp), dtype=torch.float32, device=q.device)
    if not causal and band is None:
        band = document_band(T, batch=B, device=q.device)
    lo, hi = _band_ptrs(band, B, T, q.device)
    with _lib.device_ctx(q):
        rc = _lib.lib().uamd_attn_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(do),
                                      _lib.ptr(lse), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(delta),
                                      _strides(q, k, v, o, do, dq, dk, dv), B, T, Hq, Hk, D, Tp, float(scale),
                                      1 if causal else 0, lo, hi, _lib.dtype_code(q.dtype), _lib.stream_of(q))
    _lib.check(rc, "uamd_attn_bwd")
    return dq, dk, dv


class FlashAttention(torch.autograd.Function):
    """o = causal_attention(q, k, v) on [B,T,H,D] views; saves (q, k, v, o, lse) like flash-attention."""

    @staticmethod
    def forward(ctx, q, k, v, scale, band, causal=True):
        # (causal only when it is not the default: tests and tools swap attn_forward / attn_backward for five-argument stand-ins)
        ctx.scale, ctx.band, ctx.causal = scale, band, causal
        if q.is_cuda and not native(q, k, v) and supported(q, k, v):
            # a shape that runs zero-padded (the ViT's head_dim 80, TinyLlama's 64): keep the PADDED operands for the
            # backward instead of padding q, k, v, o and the LSE a second time there (288 GB: the copies are cheaper kept than redone)
            kept = []
            o, _ = attn_forward(q, k, v, scale, band, causal, keep_padded=kept)
            ctx.save_for_backward(*kept)
            ctx.shapes = (q.shape, k.shape)
            return o
        o, lse = attn_forward(q, k, v, scale, band) if causal else attn_forward(q, k, v, scale, band, False)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.shapes = None
        return o

    @staticmethod
    def backward(ctx, do):
        if ctx.shapes is not None:
            qs, ks = ctx.shapes
            q = do.new_empty(1).expand(qs)                         # shapes only (no storage): the padded operands are what is read
            k = do.new_empty(1).expand(ks)
            dq, dk, dv = attn_backward(do, q, k, k, None, None, ctx.scale, ctx.band, ctx.causal, padded=tuple(ctx.saved_tensors))
            return dq, dk, dv, None, None, None
        q, k, v, o, lse = ctx.saved_tensors
        if ctx.causal:
            dq, dk, dv = attn_backward(do, q, k, v, o, lse, ctx.scale, ctx.band)
        else:
            dq, dk, dv = attn_backward(do, q, k, v, o, lse, ctx.scale, ctx.band, False)
        return dq, dk, dv, None, None, None


def flash_attention(q, k, v, scale=None, band=None, causal=True):
    return FlashAttention.apply(q, k, v, scale, band, causal)
