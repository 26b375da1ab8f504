"""Single-token decode launchers (csrc/decode.hip); mirror of the decode half of unsloth/kernels/utils.py.

`fast_gemv` / the bsz == 1 branch of `fast_linear_forward` (utils.py:872-977, :1082-1125) become ONE `uamd_gemv`
launch per group of projections that share the token (q|k|v, gate|up), plus one small `uamd_gemv` over the LoRA A rows
that share it. NF4 weights are read in bitsandbytes' format straight from the `Params4bit` storage: nothing is
dequantised to memory.
"""
import ctypes

import torch

from .. import _lib
from .._lib import GemvGroup, GemvPrologue

MAX_GROUPS = 4
_SYNC = {}


class HandOff:
    """Device workspace + tag source of the in-launch hand-offs (uamd_gemv_fused's t = A x, uamd_attn_decode_fused's partials).
    Every launch on a workspace needs a tag no earlier launch on it used:
      * eager launches: `next_tag()`, a host counter (step_dev None);
      * launches captured in a hipGraph: a per-launch-SITE constant `site` (1 .. TAG_STRIDE - 1) plus TAG_STRIDE x the value of
        the device counter `step_dev` (int32 [1]), which the owner advances once per replay (DecodeEngine: with the position).
    Launches that share a HandOff must be ordered (one stream / one graph)."""

    def __init__(self, dev, nbytes=None, step_dev=None):
        self.ws = torch.zeros(((nbytes or _lib.GEMV_SYNC_BYTES) + 3) // 4, dtype=torch.int32, device=dev)
        self.step_dev = step_dev
        self._host = 0

    def next_tag(self):
        # below 2^31 so that host tags and step tags (step * TAG_STRIDE + site, step >= 1) of one workspace cannot meet early; on
        # wrap-around the workspace is cleared
        self._host += 1
        if self._host >= (1 << 31):
            self.ws.zero_()
            self._host = 1
        return self._host

    def tags(self, site=None):
        """(tag, tag_dev pointer or None) for one launch."""
        if site is None or self.step_dev is None:
            return self.next_tag(), None
        assert 0 < site < _lib.TAG_STRIDE
        return int(site), self.step_dev


def sync_workspace(dev):
    """The default HandOff of a device, for eager launches on the current stream; a caller that decodes on several streams at
    once, or under a hipGraph, owns its own (`fused["sync"]`, as DecodeEngine does)."""
    key = (dev.type, dev.index)
    ws = _SYNC.get(key)
    if ws is None:
        ws = _SYNC[key] = HandOff(dev)
    return ws


def _nf4_fields(qs):
    """(absmax_u8, absmax_f32, code2, absmax2, offset, blocksize2) of a quant state; python floats cached on it."""
    if getattr(qs, "quant_type", "nf4") not in ("nf4", None):
        raise NotImplementedError(f"uamd_gemv decodes NF4 only (quant_type {qs.quant_type!r})")
    if qs.nested:
        off = getattr(qs, "_offset_float", None)
        if off is None:
            off = qs._offset_float = float(qs.offset)          # one host sync per weight, outside any graph capture
        return qs.absmax, None, qs.state2.code, qs.state2.absmax, off, int(qs.state2.blocksize)
    return None, qs.absmax, None, None, 0.0, 0


def lora_a_rows(A_list, dtype):
    """[sum R_i, K] activation-dtype copy of the LoRA A factors that share an input, cached on the first Parameter
    (the reference caches `lora_A._fast_lora = lora_A.to(dtype)`, utils.py:1104-1106; weights are frozen at inference)."""
    from .utils import _CAST_EPOCH
    key = A_list[0]
    ent = getattr(key, "_uamd_fast_lora_rows", None)
    # validity = the rule kernels/utils._cached_cast uses: _version alone is not enough -- the fused optimizers
    # (optim.FlatAdamW's raw HIP kernel, torch's fused AdamW) update parameters WITHOUT bumping it, so the global
    # optimizer-step / model-forward epoch and the storage address are part of the key (generate -> train -> generate
    # must not multiply the trained B by the pre-training A)
    vers = (_CAST_EPOCH[0],) + tuple((a._version, a.data_ptr()) for a in A_list)
    if ent is None or ent[0] != vers or ent[1].dtype != dtype or len(ent[2]) != len(A_list):
        shapes = [a.shape[0] for a in A_list]
        if (ent is not None and ent[1].dtype == dtype and ent[2] == shapes and ent[1].shape[1] == A_list[0].shape[1]
                and ent[1].device == A_list[0].device):
            # refresh IN PLACE: a captured hipGraph of the token step holds this buffer's address
            r0 = 0
            with torch.no_grad():
                for a in A_list:
                    ent[1][r0:r0 + a.shape[0]].copy_(a.detach())
                    r0 += a.shape[0]
            ent = (vers, ent[1], shapes)
        else:
            rows = torch.cat([a.detach().to(dtype) for a in A_list], dim=0).contiguous()
            ent = (vers, rows, shapes)
        key._uamd_fast_lora_rows = ent
    return ent[1]


def gemv(x, groups, nf4, blocksize=64, pro=None):
    """One launch. x: [K] (contiguous, 16-bit). groups: list of dicts with W, N and optionally qs (quant state),
    y, y_f32, lora_t, lora_b, lora_scale, R, bias. Returns the list of outputs.
    `pro` (uamd_gemv_fused): dict(mode=0|1|2, x2=, res=, norm_w=, eps=, h_out=, a_rows=, t_off=[per group], sync=HandOff, site=,
    glu=) -- the
    token is produced inside the launch (SwiGLU of x and x2 / residual add + RMSNorm; x may be None in mode 2), the LoRA
    t = A x is computed once by the launch's first workgroups from the stacked A rows (handed over through `sync`) instead
    of arriving from a launch of its own, and with glu=True the two groups (gate, up) leave ONE output h = SwiGLU(gate, up)."""
    ref = x if x is not None else pro["res"]
    _lib.require_gpu(ref)
    K = ref.numel()
    dtype, dev = ref.dtype, ref.device
    assert 1 <= len(groups) <= MAX_GROUPS
    arr = (GemvGroup * len(groups))()
    outs, keep = [], []
    for i, g in enumerate(groups):
        N = g["N"]
        y_f32 = bool(g.get("y_f32", False))
        y = g.get("y")
        if y is None:
            y = torch.empty(N, dtype=torch.float32 if y_f32 else dtype, device=dev)
        outs.append(y)
        W = g["W"]
        e = arr[i]
        e.W = W.data_ptr()
        if nf4:
            a_u8, a_f32, code2, absmax2, off, bs2 = _nf4_fields(g["qs"])
            e.absmax_u8 = a_u8.data_ptr() if a_u8 is not None else None
            e.absmax_f32 = a_f32.data_ptr() if a_f32 is not None else None
            e.code2 = code2.data_ptr() if code2 is not None else None
            e.absmax2 = absmax2.data_ptr() if absmax2 is not None else None
            e.offset, e.blocksize2, e.ldw = off, bs2, 0
        else:
            assert W.dim() == 2 and W.stride(1) == 1 and W.dtype == dtype and W.shape[1] == K
            e.ldw = W.stride(0)
        e.y = y.data_ptr()
        t = g.get("lora_t")
        if t is not None or (pro is not None and pro.get("a_rows") is not None and g.get("lora_b") is not None):
            B = g["lora_b"]
            assert (t is None or t.dtype == torch.float32) and B.stride(1) == 1
            e.lora_t = t.data_ptr() if t is not None else None
            e.lora_b, e.ld_lb = B.data_ptr(), B.stride(0)
            e.lora_scale, e.R = float(g["lora_scale"]), int(g["R"])
            e.lora_b_f32 = int(B.dtype == torch.float32)
            assert B.dtype in (torch.float32, dtype)
            keep += [t, B]
        bias = g.get("bias")
        e.bias = bias.data_ptr() if bias is not None else None
        e.N, e.y_f32 = N, int(y_f32)
    if pro is None:
        with _lib.device_ctx(ref):
            rc = _lib.lib().uamd_gemv(_lib.ptr(x), K, arr, len(groups), int(bool(nf4)), int(blocksize),
                                      _lib.dtype_code(dtype), _lib.stream_of(ref))
        _lib.check(rc, "uamd_gemv")
        return outs
    P = GemvPrologue()
    P.mode = int(pro.get("mode", 0))
    P.glu = int(bool(pro.get("glu", False)))
    for name in ("x2", "res", "norm_w", "h_out", "a_rows"):
        tns = pro.get(name)
        setattr(P, name, tns.data_ptr() if tns is not None else None)
        keep.append(tns)
    if P.mode == 1:
        assert pro["x2"].dtype == dtype and pro["x2"].numel() == K and pro["x2"].is_contiguous()
    if P.mode == 2:
        w = pro["norm_w"]
        assert w.numel() == K and w.is_contiguous() and w.dtype in (dtype, torch.float32) and pro["res"].is_contiguous()
        P.w_f32 = int(w.dtype == torch.float32 and dtype != torch.float32)
        P.eps = float(pro["eps"])
    a_rows = pro.get("a_rows")
    if a_rows is not None:
        assert a_rows.dtype == dtype and a_rows.stride(1) == 1 and a_rows.shape[1] == K
        P.Rt, P.ld_a = int(a_rows.shape[0]), int(a_rows.stride(0))
        for i, off in enumerate(pro["t_off"]):
            P.t_off[i] = int(off)
        ho = pro.get("sync")
        if ho is None:
            ho = sync_workspace(dev)
        assert isinstance(ho, HandOff) and ho.ws.numel() * 4 >= _lib.GEMV_SYNC_BYTES and ho.ws.device == dev
        tag, tag_dev = ho.tags(pro.get("site"))
        P.sync, P.tag = ho.ws.data_ptr(), tag
        P.tag_dev = tag_dev.data_ptr() if tag_dev is not None else None
        keep += [ho.ws, tag_dev]
    with _lib.device_ctx(ref):
        rc = _lib.lib().uamd_gemv_fused(_lib.ptr(x) if x is not None else None, K, arr, len(groups), int(bool(nf4)),
                                        int(blocksize), _lib.dtype_code(dtype), _lib.stream_of(ref), ctypes.byref(P))
    _lib.check(rc, "uamd_gemv_fused")
    return outs


def linear_group(x, projs, out=None, fused=None):
    """y_i = W_i x + s_i B_i (A_i x) (+ bias_i) for projections sharing the token x [K]: at most two launches (the A rows
    of all members, then the weights). projs: (W, quant_state, A, B, s[, bias]) as get_lora_parameters(_bias) returns
    them. `out`: optional preallocated [sum N_i] row the outputs are written into back to back. Returns the views.
    `fused` (dict: mode / x2 / res / norm_w / eps / h_out / sync / site / glu, see gemv): ONE launch -- the token is produced inside it
    (x may be None in mode 2), the A x products are computed by the launch's own first workgroups, and with glu=True the two
    projections (gate, up) return ONE vector h = SwiGLU(gate, up) of N entries."""
    if fused is not None:
        return _linear_group_fused(x, projs, out, fused)
    x = x.reshape(-1)
    dtype = x.dtype
    nf4 = projs[0][1] is not None
    assert all((p[1] is not None) == nf4 for p in projs), "a launch is all-NF4 or all-16-bit"
    Ns = [int(p[1].shape[0]) if p[1] is not None else int(p[0].shape[0]) for p in projs]
    if out is None:
        out = torch.empty(sum(Ns), dtype=dtype, device=x.device)
    with_lora = [p for p in projs if p[2] is not None]
    t_all = None
    if with_lora:
        A_rows = lora_a_rows([p[2] for p in with_lora], dtype)
        (t_all,) = gemv(x, [dict(W=A_rows, N=A_rows.shape[0], y_f32=True)], nf4=False)
    groups, col, r0 = [], 0, 0
    for p, N in zip(projs, Ns):
        W, qs, A, B, s = p[:5]
        g = dict(W=W, N=N, qs=qs, y=out[col:col + N], bias=p[5] if len(p) > 5 else None)
        if A is not None:
            R = A.shape[0]
            g.update(lora_t=t_all[r0:r0 + R], lora_b=B.detach(), lora_scale=s, R=R)
            r0 += R
        groups.append(g)
        col += N
    blocksize = int(projs[0][1].blocksize) if nf4 else 64
    ys = []
    for i in range(0, len(groups), MAX_GROUPS):
        ys += gemv(x, groups[i:i + MAX_GROUPS], nf4=nf4, blocksize=blocksize)
    return ys


def _linear_group_fused(x, projs, out, fused):
    ref = x if x is not None else fused["res"]
    if x is not None:
        x = x.reshape(-1)
    dtype, dev = ref.dtype, ref.device
    nf4 = projs[0][1] is not None
    assert all((p[1] is not None) == nf4 for p in projs), "a launch is all-NF4 or all-16-bit"
    assert len(projs) <= MAX_GROUPS
    Ns = [int(p[1].shape[0]) if p[1] is not None else int(p[0].shape[0]) for p in projs]
    glu = bool(fused.get("glu", False))
    if glu:
        assert len(projs) == 2 and Ns[0] == Ns[1], "glu: gate and up of one MLP"
    if out is None:
        out = torch.empty(Ns[0] if glu else sum(Ns), dtype=dtype, device=dev)
    with_lora = [p for p in projs if p[2] is not None]
    pro = dict(fused)
    pro.setdefault("mode", 0)
    t_off = [0] * len(projs)
    if with_lora:
        pro["a_rows"] = lora_a_rows([p[2] for p in with_lora], dtype)
    groups, col, r0 = [], 0, 0
    for i, (p, N) in enumerate(zip(projs, Ns)):
        W, qs, A, B, s = p[:5]
        g = dict(W=W, N=N, qs=qs, y=out[:N] if glu else out[col:col + N], bias=p[5] if len(p) > 5 else None)
        if A is not None:
            R = A.shape[0]
            g.update(lora_b=B.detach(), lora_scale=s, R=R)
            t_off[i] = r0
            r0 += R
        groups.append(g)
        col += N
    pro["t_off"] = t_off
    blocksize = int(projs[0][1].blocksize) if nf4 else 64
    ys = gemv(x, groups, nf4=nf4, blocksize=blocksize, pro=pro)
    return ys[:1] if glu else ys


def rope_kv_append(qkv, cos, sin, kv_len, k_cache, v_cache, Hq, Hk, D, rope_pos=None):
    """In place on qkv [B, (Hq + 2 Hk) D]; k_cache / v_cache [B, Hk, S_max, D]; kv_len int32 [B] on the device."""
    B = qkv.shape[0]
    assert qkv.stride(1) == 1 and k_cache.is_contiguous() and v_cache.is_contiguous() and kv_len.dtype == torch.int32
    assert cos.stride(1) == 1 and sin.stride() == cos.stride() and cos.dtype == qkv.dtype
    with _lib.device_ctx(qkv):
        rc = _lib.lib().uamd_rope_kv_append(
            _lib.ptr(qkv), qkv.stride(0), _lib.ptr(cos), _lib.ptr(sin), cos.stride(0), _lib.ptr(kv_len),
            _lib.ptr(rope_pos) if rope_pos is not None else None, _lib.ptr(k_cache), _lib.ptr(v_cache),
            k_cache.stride(0), k_cache.stride(1), B, Hq, Hk, D, k_cache.shape[2], _lib.dtype_code(qkv.dtype),
            _lib.stream_of(qkv))
    _lib.check(rc, "uamd_rope_kv_append")


def attn_decode(q, k_cache, v_cache, kv_len, out, partials, split_keys, scale, len_add=1, window=0):
    """q [B, Hq*D] (a view into the fused qkv row is fine), out [B, Hq*D]; partials fp32 [B, Hq, nsplit, D + 2]."""
    B, Hk, S_max, D = k_cache.shape
    Hq = partials.shape[1]
    nsplit = partials.shape[2]
    assert nsplit * split_keys >= S_max and q.stride(1) == 1 and out.stride(1) == 1
    with _lib.device_ctx(q):
        rc = _lib.lib().uamd_attn_decode(
            _lib.ptr(q), q.stride(0), _lib.ptr(k_cache), _lib.ptr(v_cache), k_cache.stride(0), k_cache.stride(1),
            _lib.ptr(kv_len), int(len_add), _lib.ptr(partials), _lib.ptr(out), out.stride(0), B, Hq, Hk, D, nsplit,
            int(split_keys), int(window), float(scale), _lib.dtype_code(q.dtype), _lib.stream_of(q))
    _lib.check(rc, "uamd_attn_decode")
    return out


def fused_attn_workspace(B, Hq, Hk, S_max, D, split_keys, device, step_dev=None):
    """(HandOff holding the partials as granules, counters) of uamd_attn_decode_fused for a cache of S_max positions, zeroed
    here, once."""
    nsplit = (S_max + split_keys - 1) // split_keys
    ho = HandOff(device, nbytes=B * Hq * nsplit * (D + 2) * 8, step_dev=step_dev)
    ho.nsplit = nsplit
    return ho, torch.zeros(B * Hk, dtype=torch.int32, device=device)


def attn_decode_fused(qkv, cos, sin, kv_len, k_cache, v_cache, out, partials, counters, split_keys, scale, Hq, window=0,
                      rope_pos=None, site=None):
    """RoPE on the new token's q / k, append of k / v at kv_len[b], split-KV attention over kv_len[b] + 1 keys and the
    combine, ONE launch (uamd_attn_decode_fused). qkv [B, (Hq + 2 Hk) D] is the raw q|k|v row and is left untouched; out
    [B, Hq * D]; (partials, counters) from `fused_attn_workspace`, owned by ONE stream of launches; `site`: the launch-site
    constant when the launch is captured in a hipGraph (HandOff)."""
    B, Hk, S_max, D = k_cache.shape
    nsplit = partials.nsplit
    assert nsplit * split_keys >= S_max and qkv.stride(1) == 1 and out.stride(1) == 1
    assert partials.ws.numel() * 4 >= B * Hq * nsplit * (D + 2) * 8
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and kv_len.dtype == torch.int32
    assert counters.dtype == torch.int32 and counters.numel() >= B * Hk
    assert cos.stride(1) == 1 and sin.stride() == cos.stride() and cos.dtype == qkv.dtype
    tag, tag_dev = partials.tags(site)
    with _lib.device_ctx(qkv):
        rc = _lib.lib().uamd_attn_decode_fused(
            _lib.ptr(qkv), qkv.stride(0), _lib.ptr(cos), _lib.ptr(sin), cos.stride(0), _lib.ptr(kv_len),
            _lib.ptr(rope_pos) if rope_pos is not None else None, _lib.ptr(k_cache), _lib.ptr(v_cache), k_cache.stride(0),
            k_cache.stride(1), partials.ws.data_ptr(), _lib.ptr(counters), _lib.ptr(out), out.stride(0), B, Hq, Hk, D, S_max,
            nsplit, int(split_keys), int(window), float(scale), tag, tag_dev.data_ptr() if tag_dev is not None else None,
            _lib.dtype_code(qkv.dtype), _lib.stream_of(qkv))
    _lib.check(rc, "uamd_attn_decode_fused")
    return out


def argmax_f32(logits, out=None, ws=None):
    """Greedy next token: argmax over the last dimension of contiguous fp32 logits [rows, n] -> int64 [rows] (uamd_argmax_f32).
    `ws` = (float32 [rows * 64], int64 [rows * 64]) workspaces to reuse under a hipGraph."""
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.is_contiguous()
    rows, n = logits.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.long, device=logits.device)
    if ws is None:
        ws = (torch.empty(rows * 64, dtype=torch.float32, device=logits.device),
              torch.empty(rows * 64, dtype=torch.long, device=logits.device))
    with _lib.device_ctx(logits):
        rc = _lib.lib().uamd_argmax_f32(_lib.ptr(logits), rows, n, _lib.ptr(ws[0]), _lib.ptr(ws[1]), _lib.ptr(out),
                                        _lib.stream_of(logits))
    _lib.check(rc, "uamd_argmax_f32")
    return out
