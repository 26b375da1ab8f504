"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain torch/numpy restatement, on the CPU, of the arithmetic of the reference's hot path
(unslothai/unsloth, Triton kernels + manual autograd), one function per reference kernel, each
citing the reference file:line it follows. Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this package; the product (`unsloth_amd/`) never does.

Every function computes in fp32 and rounds to the activation dtype exactly where the reference
rounds ("rounding points", SURVEY 7 "Hard parts"):
  * RMSNorm: x_hat rounded to W.dtype before * W, the product taken in W.dtype  (rms_layernorm.py:56-58)
  * SwiGLU/GeGLU: f rounded to the activation dtype before * g; h, df, dg are products IN that dtype;
    de is fp32 math on the rounded dg, rounded once                          (swiglu.py:37-47, 92-104)
  * RoPE: when Q and the table share a 16-bit dtype, q*cos, q*sin and their sum/difference are each
    rounded (Triton computes them in that dtype)                             (rope_embedding.py:77-89)
  * CE: everything fp32, gradient rounded once into the logits dtype         (cross_entropy_loss.py:231-276)

PINNING: the fp32 and fp16 behaviour of these functions is pinned against the reference's OWN Triton
kernels executed under TRITON_INTERPRET=1 (oracle/make_golden_from_reference.py ->
tests/golden/ref_triton_*.pt, checked by tests/test_oracle_golden.py). bf16 cannot run under the
interpreter (numpy has no bf16), so the bf16 rounding points are pinned only by construction
(same code path as fp16 with the dtype swapped). The NF4 format (third-party bitsandbytes) and the
fused linear-CE (third-party unsloth_zoo) are PARITY-UNPINNED: no vectors exist in the reference.
"""
import math

import numpy as np
import torch

F32 = torch.float32


def rt(x, dtype):
    """round-trip through dtype: the value the reference would hold after `.to(dtype)`."""
    return x.to(dtype).to(F32)


# ------------------------------------------------------------------------------------------------
# RMSNorm                                                     unsloth/kernels/rms_layernorm.py
def rms_layernorm_forward(X, W, eps, gemma=False):
    """:40-59 (and :141-159 for gemma). Returns (Y in X.dtype, r fp32 [rows])."""
    shape = X.shape
    x = X.reshape(-1, shape[-1]).to(F32)
    row_var = (x * x).sum(dim=1) / x.shape[1]
    r = torch.rsqrt(row_var + eps)
    normed = x * r[:, None]
    if gemma:
        y = normed * (W.to(F32) + 1.0)
    else:
        normed = rt(normed, W.dtype)                 # normed.to(W_row.dtype)
        y = rt(normed * W.to(F32), W.dtype)           # product in W's dtype
    return y.to(X.dtype).view(shape), r


def rms_layernorm_backward(dY, X, W, r, gemma=False):
    """:84-112. dX in dY.dtype."""
    shape = dY.shape
    n = shape[-1]
    dy = dY.reshape(-1, n).to(F32)
    x = X.reshape(-1, n).to(F32)
    w = W.to(F32)
    normed = x * r[:, None]
    dyw = dy * (w + 1.0) if gemma else dy * w
    rowsum = (dyw * normed).sum(dim=1, keepdim=True)
    out = r[:, None] / n * (n * dyw - normed * rowsum)
    return out.to(dY.dtype).view(shape)


# ------------------------------------------------------------------------------------------------
# RoPE                                                        unsloth/kernels/rope_embedding.py
def _rope_rotate(q0, q1, c, s, q_dtype, t_dtype):
    native = q_dtype == t_dtype and q_dtype in (torch.float16, torch.bfloat16)
    if native:   # every product and the sum are rounded to the shared 16-bit dtype
        o0 = rt(q0 * c, q_dtype) - rt(q1 * s, q_dtype)
        o1 = rt(q1 * c, q_dtype) + rt(q0 * s, q_dtype)
    else:        # promoted to fp32, one rounding on store
        o0 = q0 * c - q1 * s
        o1 = q1 * c + q0 * s
    return o0.to(q_dtype), o1.to(q_dtype)


def rope_embedding_qk(Q, K, cos, sin, rope_indices=None, backward=False):
    """:23-98. Q [B,Hq,T,D], K [B,Hk,T,D] (or None); position = indices[b*T+t] or t. Out of place."""
    B, Hq, T, D = Q.shape
    half = D // 2
    cos = cos.squeeze()
    sin = sin.squeeze()
    if rope_indices is not None:
        pos = rope_indices.reshape(-1).to(torch.int64)                    # :46-52
    else:
        pos = torch.arange(B * T, dtype=torch.int64) % T                   # :54
    c = cos[pos, :half].to(F32).view(B, 1, T, half)
    s = sin[pos, :half].to(F32).view(B, 1, T, half)
    if backward:
        s = -s                                                             # :71-72
    outs = []
    for X in (Q, K):
        if X is None:
            outs.append(None)
            continue
        x = X.to(F32)
        o0, o1 = _rope_rotate(x[..., :half], x[..., half:], c, s, X.dtype, cos.dtype)
        outs.append(torch.cat([o0, o1], dim=-1))
    return outs[0], outs[1]


def rope_embedding_dense(Q, cos, sin, backward=False):
    """:104-166 via Fast_RoPE_Embedding (:169-261): Q [B,T,H,D], position = row % seqlen.
    NOTE the dense kernel casts Q to the table dtype first (:153-155)."""
    B, T, H, D = Q.shape
    q = Q.permute(0, 2, 1, 3)
    out, _ = rope_embedding_qk(q, None, cos, sin, None, backward)
    return out.permute(0, 2, 1, 3).contiguous()


# ------------------------------------------------------------------------------------------------
# SwiGLU / GeGLU                              unsloth/kernels/swiglu.py, unsloth/kernels/geglu.py
def _act(e, kind):
    """returns (f, df/de) in fp32 for e fp32."""
    if kind == "swiglu":
        se = torch.sigmoid(e)
        return e * se, se * (1.0 + e * (1.0 - se))                         # swiglu.py:41, 103
    if kind == "geglu_exact":
        fp = 0.5 * (torch.erf(e * (1.0 / math.sqrt(2.0))) + 1.0)           # geglu.py:100
        t = 0.3989422804014327
        return fp * e, fp + t * e * torch.exp(-0.5 * e * e)                # geglu.py:113
    if kind == "geglu_approx":
        s = 0.7978845608028654
        a = s * e
        b = a * 0.044715 * e * e
        T = 1.0 + torch.tanh(a + b)
        T2 = 0.5 * T
        Q2 = -T2 * (T - 2.0) * (a + 3.0 * b)                               # geglu.py:221-225
        return T2 * e, T2 + Q2
    raise ValueError(kind)


def glu_forward(e, g, kind="swiglu"):
    """swiglu.py:37-47 / geglu.py:43-53 / :154-167: h = f(e).to(dtype) * g, product in dtype."""
    dt = e.dtype
    f, _ = _act(e.to(F32), kind)
    return rt(rt(f, dt) * g.to(F32), dt).to(dt)


def glu_backward(DW, e, g, kind="swiglu"):
    """swiglu.py:86-109 / geglu.py:95-123 / :218-244. Returns (h, df, de) -- the values the
    reference leaves in (DW, e, g)."""
    dt = e.dtype
    ef = e.to(F32)
    f32, dfde = _act(ef, kind)
    f = rt(f32, dt)
    dw, gf = DW.to(F32), g.to(F32)
    h = rt(f * gf, dt)
    df = rt(dw * f, dt)
    dg = rt(dw * gf, dt)
    if kind == "swiglu":
        se = torch.sigmoid(ef)
        de = dg * se * (1.0 + ef * (1.0 - se))
    else:
        de = dg * dfde
    return h.to(dt), df.to(dt), de.to(dt)


# ------------------------------------------------------------------------------------------------
# Cross entropy                                        unsloth/kernels/cross_entropy_loss.py
def _ce_transform(x, softcap, scale):
    if scale:
        x = scale * x                                                      # :79-80
    if softcap:
        x = softcap * torch.tanh(x / softcap)                              # :82-83
    return x


def cross_entropy_forward(logits, labels, softcap=0.0, scale=0.0):
    """:68-102 (+ :152-190 and the host reduction :366-370, same quantity). logits [rows, V].
    Returns (loss fp32 [rows], logsumexp fp32 [rows])."""
    x = _ce_transform(logits.to(F32), softcap, scale)
    c = x.max(dim=1).values
    lse = c + torch.log(torch.exp(x - c[:, None]).sum(dim=1))
    lab = labels.to(torch.int64)
    valid = lab != -100
    xl = x.gather(1, lab.clamp(min=0)[:, None])[:, 0]
    loss = torch.where(valid, lse - xl, torch.zeros_like(lse))
    return loss, lse


def cross_entropy_backward(logits, dloss, lse, labels, softcap=0.0, scale=0.0):
    """:231-276. Returns the gradient in logits.dtype (what the reference writes over logits)."""
    x = logits.to(F32)
    if scale:
        x = x * scale
    partial = x
    if softcap:
        partial = torch.tanh(x / softcap)
        x = softcap * partial
    y = torch.exp(x - lse[:, None])
    lab = labels.to(torch.int64)
    onehot = torch.zeros_like(y)
    valid = lab != -100
    onehot[valid, lab[valid]] = 1.0
    y = y - onehot
    if scale:
        y = y * scale
    if softcap:
        y = y * (1.0 - partial * partial)
    dl = torch.where(valid, dloss.to(F32), torch.zeros_like(dloss, dtype=F32))
    return (dl[:, None] * y).to(logits.dtype)


def fast_cross_entropy_loss(logits, labels, softcap=0.0, scale=0.0, n_items=None):
    """:421-449: sum / n_items."""
    B, T, V = logits.shape
    loss, _ = cross_entropy_forward(logits.view(B * T, V), labels.view(-1), softcap, scale)
    if n_items is None:
        n_items = torch.count_nonzero(labels != -100)
    return loss.sum() / n_items


def shift_labels(labels):
    """unsloth/models/llama.py:1545-1551."""
    out = torch.empty_like(labels)
    out[..., :-1] = labels[..., 1:]
    out[..., -1] = -100
    return out


def fused_linear_ce(hidden, weight, labels, n_items=None, softcap=0.0, scale=0.0):
    """Oracle for unsloth_fused_ce_loss (third party unsloth_zoo, call site llama.py:1497-1509):
    F.cross_entropy(lm_head(h).float()[..., :-1, :], labels[..., 1:], sum) / n_items, which the
    materialised-logits branch (llama.py:1525-1562) is numerically equivalent to. The logits pass
    through the activation dtype (matmul output) before the fp32 CE. Returns (loss, d_hidden)."""
    h = hidden.detach().to(F32).requires_grad_(True)
    logits_f = h.reshape(-1, h.shape[-1]) @ weight.to(F32).t()
    # value rounded through the activation dtype, gradient straight through the rounding
    logits = logits_f + (rt(logits_f.detach(), hidden.dtype) - logits_f.detach())
    lab = shift_labels(labels).reshape(-1)
    x = _ce_transform(logits, softcap, scale)
    if n_items is None:
        n_items = torch.count_nonzero(lab != -100)
    loss = torch.nn.functional.cross_entropy(x, lab, ignore_index=-100, reduction="sum") / n_items
    (dh,) = torch.autograd.grad(loss, h)
    return loss.detach(), dh.to(hidden.dtype)


# ------------------------------------------------------------------------------------------------
# NF4 (bitsandbytes format, third party -- restated from the published algorithm)
NF4_CODE = np.array([
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
    0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0], dtype=np.float32)


def nf4_quantize_np(w, blocksize=64):
    """First-level NF4: absmax per block, x * (1/absmax) and the nearest code with strict '>'
    midpoint boundaries (bitsandbytes dQuantizeNF4). w: float32 numpy, size % blocksize == 0.
    Returns (packed uint8 [n/2], absmax float32 [n/blocksize])."""
    w = np.asarray(w, dtype=np.float32).reshape(-1, blocksize)
    absmax = np.abs(w).max(axis=1).astype(np.float32)
    inv = np.float32(1.0) / np.where(absmax > 0, absmax, np.float32(1.0))
    x = (w * inv[:, None]).astype(np.float32)
    thr = (np.float32(0.5) * (NF4_CODE[:-1] + NF4_CODE[1:])).astype(np.float32)
    codes = (x[..., None] > thr).sum(axis=-1).astype(np.uint8)
    codes[absmax == 0] = 7
    codes = codes.reshape(-1)
    packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)      # high nibble = even element
    return packed, absmax


def nf4_dequantize_np(packed, absmax_f32, blocksize=64, lut=None):
    """W[j] = LUT[code_j] * absmax[j // blocksize], fp32 product (unsloth/kernels/utils.py:662-675
    -> bitsandbytes kDequantizeBlockwise NF4)."""
    lut = NF4_CODE if lut is None else np.asarray(lut, dtype=np.float32)
    packed = np.asarray(packed, dtype=np.uint8).reshape(-1)
    codes = np.empty(packed.size * 2, dtype=np.uint8)
    codes[0::2] = packed >> 4
    codes[1::2] = packed & 15
    scale = np.repeat(np.asarray(absmax_f32, dtype=np.float32), blocksize)[: codes.size]
    return (lut[codes] * scale).astype(np.float32)


def dequantize_absmax_np(absmax_u8, code2, absmax2, offset, blocksize2=256):
    """absmax_f32[k] = code2[absmax_u8[k]] * absmax2[k // blocksize2] + offset
    (unsloth/kernels/utils.py:650-659)."""
    a = np.asarray(code2, dtype=np.float32)[np.asarray(absmax_u8, dtype=np.uint8)]
    s = np.repeat(np.asarray(absmax2, dtype=np.float32), blocksize2)[: a.size]
    return (a * s + np.float32(offset)).astype(np.float32)


def nf4_dequantize_state(packed, qs, dtype=None):
    """Dequantise with a (possibly nested) quant state object exposing the bitsandbytes attribute
    names; returns a torch tensor [out, in] in qs.dtype."""
    to_np = lambda t: t.detach().cpu().numpy()
    if getattr(qs, "nested", False) or getattr(qs, "state2", None) is not None:
        am = dequantize_absmax_np(to_np(qs.absmax), to_np(qs.state2.code), to_np(qs.state2.absmax),
                                  float(qs.offset), qs.state2.blocksize)
    else:
        am = to_np(qs.absmax).astype(np.float32)
    w = nf4_dequantize_np(to_np(packed), am, qs.blocksize, to_np(qs.code) if qs.code is not None else None)
    return torch.from_numpy(w).view(*qs.shape).to(dtype or qs.dtype)


def nf4_fp32_weight(packed, qs):
    """The fp32 value of every NF4 code (LUT * absmax in fp32, no rounding to the activation dtype): the exact weight
    the decode GEMV is measured against."""
    to_np = lambda t: t.detach().cpu().numpy()
    if getattr(qs, "nested", False) or getattr(qs, "state2", None) is not None:
        am = dequantize_absmax_np(to_np(qs.absmax), to_np(qs.state2.code), to_np(qs.state2.absmax),
                                  float(qs.offset), qs.state2.blocksize)
    else:
        am = to_np(qs.absmax).astype(np.float32)
    return torch.from_numpy(nf4_dequantize_np(to_np(packed), am, qs.blocksize)).view(*qs.shape)


def gemv_4bit_naive(x, W32_codes_lut, absmax_rows, dtype, blocksize=64):
    """Rounding points of bitsandbytes' kgemm_4bit_inference_naive, the kernel the reference's fast_gemv calls
    (unsloth/kernels/utils.py:953-974; bitsandbytes >= 0.45.5 csrc/kernels.cu -- third party, source absent here,
    restated from the published kernel: PARITY UNPINNED). Per output row: quant_map = T(NF4 code), local_absmax =
    T(absmax), local_B = T(quant_map * local_absmax), local_C += float(T(local_A * local_B)) accumulated in fp32, the
    row result rounded to T. `W32_codes_lut` = NF4 code value per weight (fp32 [N, K]), `absmax_rows` fp32 [N, K/blocksize],
    x [K] in T. Returns fp32 [N] (the T-rounded outputs)."""
    T = dtype
    qm = W32_codes_lut.to(T)                                        # quant_map[i] = T(datatype[i])
    am = absmax_rows.to(T).repeat_interleave(blocksize, dim=1)       # local_absmax = T(absmax)
    B = (qm.float() * am.float()).to(T)                              # product in T
    prod = (x.to(T).float()[None, :] * B.float()).to(T)              # local_A * local_B in T
    return prod.float().sum(dim=1).to(T).float()


# ------------------------------------------------------------------------------------------------
# LoRA linear algebra                  unsloth/kernels/utils.py:1128-1170, unsloth/kernels/fast_lora.py
def matmul_lora(X, W, A, B, s):
    """out = X @ W^T ; out += (X @ A^T) @ (s * B^T)  with the reference's rounding points:
    the base product is rounded to the activation dtype (torch.matmul output), XA is rounded to the
    activation dtype, addmm_ accumulates in fp32 and rounds once (:1158-1168)."""
    dt = X.dtype
    x = X.reshape(-1, X.shape[-1]).to(F32)
    out = rt(x @ W.to(F32).t(), dt)
    if A is not None:
        xa = rt(x @ rt(A.to(F32), dt).t(), dt)
        out = rt(out + s * (xa @ rt(B.to(F32), dt).t()), dt)
    return out.to(dt).view(*X.shape[:-1], -1)


def lora_linear_grads(X, dY, W, A, B, s):
    """(dX, dA, dB) of Y = X W^T + s (X A^T) B^T in fp32 math (fast_lora.py:172-204 / :639-647):
       dX = dY W + s (dY B) A ;  dA = s (dY B)^T X ;  dB = s dY^T (X A^T)."""
    x = X.reshape(-1, X.shape[-1]).to(F32)
    dy = dY.reshape(-1, dY.shape[-1]).to(F32)
    dX = dy @ W.to(F32)
    dA = dB = None
    if A is not None:
        a, b = A.to(F32), B.to(F32)
        dX = dX + s * (dy @ b) @ a
        dA = s * (dy @ b).t() @ x
        dB = s * dy.t() @ (x @ a.t())
    return dX.view(X.shape), dA, dB


def lora_mlp_forward(X, gate, up, down, kind="swiglu"):
    """LoRA_MLP.forward (fast_lora.py:93-96). gate/up/down = (W, A, B, s). Returns (i, e, g, h)."""
    e = matmul_lora(X, *gate)
    g = matmul_lora(X, *up)
    h = glu_forward(e, g, kind)
    i = matmul_lora(h, *down)
    return i, e, g, h


def lora_mlp_reference_grads(X, gate, up, down, dY, kind="swiglu"):
    """Autograd ground truth (fp32, no intermediate rounding) for LoRA_MLP: grads w.r.t. X and the
    six LoRA matrices. Used with a tolerance, not bitwise."""
    Xf = X.detach().to(F32).requires_grad_(True)
    params = []
    outs = []
    for (W, A, B, s) in (gate, up, down):
        A = A.detach().to(F32).requires_grad_(True)
        B = B.detach().to(F32).requires_grad_(True)
        params += [A, B]
        outs.append((W.to(F32), A, B, s))
    lin = lambda x, p: x @ p[0].t() + p[3] * (x @ p[1].t()) @ p[2].t()
    e = lin(Xf, outs[0])
    g = lin(Xf, outs[1])
    if kind == "swiglu":
        f = e * torch.sigmoid(e)
    elif kind == "geglu_exact":
        f = torch.nn.functional.gelu(e)
    else:
        f = torch.nn.functional.gelu(e, approximate="tanh")
    i = lin(f * g, outs[2])
    grads = torch.autograd.grad(i, [Xf] + params, dY.to(F32))
    return i.detach(), grads


# ------------------------------------------------------------------------------------------------
# LayerNorm (unsloth/kernels/layernorm.py:25-104)
def layernorm_forward(X, W, b, eps):
    """y = ((x - mean) * rsqrt(mean((x - mean)^2) + eps)) * W + b, all in fp32, one rounding to X's dtype
    (layernorm.py:47-65). Returns (Y, r fp32 [rows], mu fp32 [rows])."""
    x = X.float()
    mu = x.mean(dim=-1, keepdim=True)
    xx = x - mu
    r = torch.rsqrt((xx * xx).mean(dim=-1, keepdim=True) + eps)
    y = (xx * r) * W.float() + b.float()
    return y.to(X.dtype), r.squeeze(-1), mu.squeeze(-1)


def layernorm_backward(dY, X, W, r, mu):
    """dX = (g - mean(g) - normed * mean(g * normed)) * r with g = dY * W, normed = (x - mu) * r; fp32, one rounding
    (layernorm.py:90-104). No dW / db (Fast_Layernorm.backward returns None for them, :163)."""
    x, g = X.float(), dY.float() * W.float()
    normed = (x - mu.unsqueeze(-1)) * r.unsqueeze(-1)
    dx = (g - g.mean(dim=-1, keepdim=True) - normed * (g * normed).mean(dim=-1, keepdim=True)) * r.unsqueeze(-1)
    return dx.to(dY.dtype)
