"""NF4 (bitsandbytes 4-bit NormalFloat) quantisation state, quantiser and 4-bit Linear, without
bitsandbytes.

The reference never touches these bytes itself: it builds a `BitsAndBytesConfig(nf4,
double_quant)` (unsloth/models/llama.py:2615-2626) and reads `W.quant_state` fields in
`fast_dequantize` (unsloth/kernels/utils.py:582-606): `absmax, shape, dtype, blocksize, offset,
state2.{absmax, code, blocksize}`. This module provides objects with exactly those attribute
names so `get_lora_parameters` / `fast_dequantize` read them the same way, plus the serialised
key layout of bitsandbytes checkpoints:
    weight, weight.absmax, weight.quant_map, weight.nested_absmax, weight.nested_quant_map,
    weight.quant_state.bitsandbytes__nf4
Format restated from the published bitsandbytes algorithm (pinned >=0.45.5 by the reference,
pyproject.toml:473); parity against bitsandbytes itself is UNPINNED (not installed here).
"""
import json

import torch

from . import _lib

NF4_CODE = [
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
    0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0,
]


def create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8):
    """bitsandbytes' 8-bit "dynamic" quantisation map (functional.create_dynamic_map), used for
    the nested quantisation of absmax. Restated; the map also travels inside the quant state, so
    kernels read it from there and never depend on this function."""
    data = []
    non_sign_bits = total_bits - 1
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    for i in range(max_exponent_bits):
        fraction_items = int(
            2 ** (i + non_sign_bits - max_exponent_bits) + 1
            if signed
            else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1
        )
        boundaries = torch.linspace(0.1, 1, fraction_items)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    if additional_items > 0:
        boundaries = torch.linspace(0.1, 1, additional_items + 1)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        if signed:
            data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    data.append(0)
    data.append(1.0)
    assert len(data) == 2**total_bits
    data.sort()
    return torch.tensor(data, dtype=torch.float32)


class QuantState:
    """Attribute-compatible with bitsandbytes.functional.QuantState."""

    def __init__(self, absmax, shape=None, code=None, blocksize=64, quant_type="nf4", dtype=None,
                 offset=None, state2=None):
        self.absmax = absmax
        self.shape = torch.Size(shape) if shape is not None else None
        self.code = code
        self.blocksize = blocksize
        self.quant_type = quant_type
        self.dtype = dtype
        self.offset = offset
        self.state2 = state2
        self.nested = state2 is not None
        self._absmax_f32 = None  # lazily de-nested statistics (288 GB HBM: keep, do not recompute)

    def to(self, device):
        self.absmax = self.absmax.to(device)
        if self.code is not None:
            self.code = self.code.to(device)
        if self.nested:
            self.offset = self.offset.to(device)
            self.state2.absmax = self.state2.absmax.to(device)
            self.state2.code = self.state2.code.to(device)
        self._absmax_f32 = None
        self._resident = None
        self._resident_group = None
        return self

    # ---- bitsandbytes checkpoint (safetensors) layout ---------------------------------------
    def as_dict(self, packed=True):
        qs = {
            "quant_type": self.quant_type, "blocksize": self.blocksize,
            "dtype": str(self.dtype).replace("torch.", ""), "shape": tuple(self.shape),
        }
        out = {"absmax": self.absmax, "quant_map": self.code}
        if self.nested:
            qs.update(nested_blocksize=self.state2.blocksize, nested_dtype="float32",
                      nested_offset=float(self.offset))
            out.update(nested_absmax=self.state2.absmax, nested_quant_map=self.state2.code)
        if packed:
            blob = torch.tensor(list(json.dumps(qs).encode("utf-8")), dtype=torch.uint8)
            out["quant_state.bitsandbytes__" + self.quant_type] = blob
        else:
            out.update(qs)
        return out

    @classmethod
    def from_dict(cls, qs_dict, device):
        qs_dict = dict(qs_dict)
        key = [k for k in qs_dict if "quant_state.bitsandbytes__" in k]
        if key:
            blob = qs_dict.pop(key[0])
            qs_dict.update(json.loads(bytes(blob.cpu().tolist()).decode("utf-8")))
        qs_dict = {k.split(".")[-1]: v for k, v in qs_dict.items()}
        state2, offset = None, None
        if "nested_absmax" in qs_dict:
            offset = torch.tensor(float(qs_dict["nested_offset"]), dtype=torch.float32, device=device)
            state2 = cls(absmax=qs_dict["nested_absmax"].to(device), blocksize=qs_dict["nested_blocksize"],
                         code=qs_dict["nested_quant_map"].to(device), dtype=torch.float32,
                         quant_type=None)
        return cls(absmax=qs_dict["absmax"].to(device), shape=qs_dict["shape"],
                   code=qs_dict["quant_map"].to(device), blocksize=qs_dict["blocksize"],
                   quant_type=qs_dict["quant_type"], dtype=getattr(torch, qs_dict["dtype"]),
                   offset=offset, state2=state2)


def absmax_f32(quant_state):
    """fp32 absmax per block, de-nesting once: code2[absmax_u8]*absmax2 + offset
    (the arithmetic of utils.py:650-659). Cached on the state: statistics are frozen."""
    qs = quant_state
    if not qs.nested:
        return qs.absmax
    if qs._absmax_f32 is None or qs._absmax_f32.device != qs.absmax.device:
        _lib.require_gpu(qs.absmax)
        n = qs.absmax.numel()
        out = torch.empty(n, dtype=torch.float32, device=qs.absmax.device)
        offset = float(qs.offset)  # one-time sync per weight (cached afterwards)
        with _lib.device_ctx(out):
            rc = _lib.lib().uamd_dequantize_absmax(
                _lib.ptr(qs.state2.code), _lib.ptr(qs.absmax), _lib.ptr(qs.state2.absmax), offset,
                _lib.ptr(out), qs.state2.blocksize, n, _lib.stream_of(out))
        _lib.check(rc, "uamd_dequantize_absmax")
        qs._absmax_f32 = out
    return qs._absmax_f32


def _quantize_blockwise_8bit(x, code, blocksize):
    """Nested statistics quantiser: nearest entry of `code` after per-block absmax scaling
    (bitsandbytes quantize_blockwise). Host logic on tiny tensors -> plain torch ops."""
    n = x.numel()
    pad = (-n) % blocksize
    xp = torch.nn.functional.pad(x.float(), (0, pad)).view(-1, blocksize)
    absmax2 = xp.abs().amax(dim=1)
    scaled = xp / absmax2.clamp_min(1e-30).unsqueeze(1)
    code = code.to(x.device)
    mids = (code[:-1] + code[1:]) / 2
    idx = torch.bucketize(scaled.reshape(-1)[:n].contiguous(), mids)
    return idx.to(torch.uint8), absmax2


def quantize_nf4(W, blocksize=64, compress_statistics=True):
    """W [out,in] (fp32/fp16/bf16, on the GPU) -> (packed uint8 [out*in/2, 1], QuantState).
    First level through the HIP kernel (uamd_nf4_quantize); double quantisation of absmax with
    blocksize 256 and the fp32 mean offset as bitsandbytes quantize_4bit does."""
    _lib.require_gpu(W)
    W = W.contiguous()
    n = W.numel()
    if n % blocksize:
        raise ValueError(f"numel {n} is not a multiple of blocksize {blocksize}")
    packed = torch.empty((n // 2, 1), dtype=torch.uint8, device=W.device)
    absmax = torch.empty(n // blocksize, dtype=torch.float32, device=W.device)
    with _lib.device_ctx(W):
        rc = _lib.lib().uamd_nf4_quantize(_lib.ptr(W), _lib.ptr(packed), _lib.ptr(absmax), n, blocksize,
                                          _lib.dtype_code(W.dtype), _lib.stream_of(W))
    _lib.check(rc, "uamd_nf4_quantize")
    code = torch.tensor(NF4_CODE, dtype=torch.float32, device=W.device)
    if compress_statistics:
        offset = absmax.mean()
        code2 = create_dynamic_map().to(W.device)
        q, absmax2 = _quantize_blockwise_8bit(absmax - offset, code2, 256)
        state2 = QuantState(absmax=absmax2, code=code2, blocksize=256, dtype=torch.float32, quant_type=None)
        qs = QuantState(absmax=q, shape=W.shape, dtype=W.dtype, blocksize=blocksize, code=code,
                        quant_type="nf4", offset=offset, state2=state2)
    else:
        qs = QuantState(absmax=absmax, shape=W.shape, dtype=W.dtype, blocksize=blocksize, code=code,
                        quant_type="nf4")
    return packed, qs


# Decoded mirrors (UNSLOTH_AMD_RESIDENT_WEIGHTS=0 | auto | 1, or nf4.set_resident(True[, model=...])): keep the DECODED bf16
# copy of an NF4 weight in HBM after its first decode instead of decoding it again at every use (2 decodes per weight per step:
# 6.9 ms of a 231 ms step at 8192 tokens, 9 % of the step at 2048). Costs 2 B/param (14 GB for Llama-3-8B's projections) of
# the 288 GB -- the NF4 bytes stay the source of truth (checkpoints, merging); frozen weights never change, so the mirror
# cannot go stale, and the GEMMs read the same bf16 values either way (bit-identical steps).
#   "0" (default; rounds 1-3 and again since round 5): never, unless a caller asks. This is what "QLoRA NF4" means in the
#        reference -- the weights are decoded inside every step -- and what bench.py's headline number pays.
#   "auto": the same fit-to-memory decision as the checkpointing schedule, taken with it and PER MODEL: the bare
#        `use_gradient_checkpointing="unsloth"` turns the mirrors of THAT model's projections on when every layer already keeps
#        everything AND the HBM left after that still holds them with room to spare (models/fast_layer.auto_policy), and off
#        again when memory gets short. A second NF4 model in the process (a reference / policy model) takes its own decision.
#   "1": every NF4 weight of the process, always (the process-wide switch: nf4.set_resident(True)).
# The switch lives on the quant states (`_mirror_on`), set for one model's projections by set_resident(on, model=m); the
# module-level RESIDENT is only the process-wide "1".
import os as _os

# Besides the mirrors there is the STEP decode (UNSLOTH_AMD_STEP_DECODE = "0" | "auto" | "1", default "0"): under the
# whole-layer Function (models/fast_layer.py) with the keep-everything policy, a layer's decoded weights live from its
# forward to its backward of the SAME step and are given back there -- every NF4 weight is decoded once per step instead of
# twice, inside the step, at the price of one decoded copy of the projections at the forward / backward turning point
# (13 GB for Llama-3-8B; "auto": part of the fit-to-memory decision, fast_layer.auto_policy). Nothing outlives the step.
# Measured at 4 x 2048 tokens: +0.5 % tokens/s for +12.4 GB of peak VRAM (the freshly allocated copies are read from HBM where
# the reused scratch of the per-use decode is read from the 256 MB cache) -- opt-in, and an `alt` point of bench.py.
STEP_DECODE_MODE = _os.environ.get("UNSLOTH_AMD_STEP_DECODE", "0")
RESIDENT_MODE = _os.environ.get("UNSLOTH_AMD_RESIDENT_WEIGHTS", "0")
RESIDENT = RESIDENT_MODE == "1"
import weakref as _weakref

_MIRRORED = _weakref.WeakSet()      # quant states that carry a decoded mirror (`_resident`, `_resident_group`)
_MARKED = _weakref.WeakSet()        # quant states whose `_mirror_on` a per-model set_resident(True, model=m) raised
_MARKED_MODELS = _weakref.WeakSet() # ... and those models (their `_uamd_mirrors_auto` record)


def _model_quant_states(model):
    return [m.weight.quant_state for m in model.modules()
            if isinstance(m, Linear4bit) and getattr(m.weight, "quant_state", None) is not None]


def mirror_wanted(qs):
    """May this weight keep a decoded mirror? The process-wide switch, or its own model's decision."""
    return RESIDENT or getattr(qs, "_mirror_on", False)


def _drop(q):
    q._resident = None
    q._resident_group = None
    _MIRRORED.discard(q)


def step_keep(qs_list):
    """The forward of a layer under the step decode: its weights keep what this forward decodes (until step_release)."""
    for q in qs_list:
        if q is not None and not mirror_wanted(q):
            q._mirror_on = True
            q._step_owned = True


def step_release(qs_list):
    """The backward of that layer has consumed them."""
    for q in qs_list:
        if q is not None and getattr(q, "_step_owned", False):
            q._step_owned = False
            q._mirror_on = False
            _drop(q)


def set_resident(on, auto=False, model=None):
    """model=None: the process-wide switch (every NF4 weight; off also clears every per-model decision). model=m: the
    projections of m only; `auto` records on m that the fit-to-memory decision (not the caller) turned them on."""
    global RESIDENT
    if model is None:
        RESIDENT = bool(on)
        if not on:
            # every per-model decision too: the states that already hold a mirror AND the ones only marked so far (a model
            # switched on with model=m whose weights have not been decoded yet), and the owners' "auto" records
            for q in list(_MIRRORED) + list(_MARKED):
                q._mirror_on = False
                _drop(q)
            _MARKED.clear()
            for m in list(_MARKED_MODELS):
                m._uamd_mirrors_auto = False
            _MARKED_MODELS.clear()
        return
    for q in _model_quant_states(model):
        q._mirror_on = bool(on)
        if on:
            _MARKED.add(q)
        else:
            _MARKED.discard(q)
            if not RESIDENT:
                _drop(q)
    model._uamd_mirrors_auto = bool(on) and bool(auto)
    if on:
        _MARKED_MODELS.add(model)
    else:
        _MARKED_MODELS.discard(model)


def mirrors_on(model):
    """Does any projection of `model` want a mirror (by its own decision or the process-wide switch)?"""
    return any(mirror_wanted(q) for q in _model_quant_states(model))


def resident_count(model=None):
    qs = _MIRRORED if model is None else _model_quant_states(model)
    return sum(1 for q in qs if getattr(q, "_resident", None) is not None)


def resident_bytes(model=None):
    qs = _MIRRORED if model is None else _model_quant_states(model)
    return sum(q._resident.numel() * q._resident.element_size() for q in qs if getattr(q, "_resident", None) is not None)


def resident_group(packed_list, qs_list):
    """Stacked row-major decode [W_1; W_2; ...] of weights that share their input (q/k/v, gate/up), decoded once and
    kept ON the quant states (the mirror lives exactly as long as the weight it mirrors). Returns
    (buffer, [row slices]); the slices also serve single-weight lookups."""
    first = qs_list[0]
    ent = getattr(first, "_resident_group", None)
    if ent is None or len(ent[0]) != len(qs_list) or any(a() is not b for a, b in zip(ent[0], qs_list)):
        cols = first.shape[1]
        rows = sum(q.shape[0] for q in qs_list)
        buf = torch.empty((rows, cols), dtype=first.dtype, device=packed_list[0].device)
        r, rows_of = 0, []
        for q in qs_list:
            rows_of.append(buf[r:r + q.shape[0]])
            r += q.shape[0]
        dequantize_nf4_group(packed_list, qs_list, rows_of)
        for q, view in zip(qs_list, rows_of):
            q._resident = view
            _MIRRORED.add(q)
        first._resident_group = ([_weakref.ref(q) for q in qs_list], buf)
        ent = first._resident_group
    buf = ent[1]
    views, r = [], 0
    for q in qs_list:
        views.append(buf[r:r + q.shape[0]])
        r += q.shape[0]
    return buf, views


_SCRATCH = {}


def scratch(device, numel, dtype, slot=0):
    """Per-device reusable buffer, the analogue of WEIGHT_BUFFERS (utils.py:608-632). The view it
    returns is overwritten by the next call with the same slot on that device (stream ordered)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), dtype, slot)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty(numel, dtype=dtype, device=device)
        _SCRATCH[key] = buf
    return buf[:numel]


def dequantize_nf4(packed, quant_state, out=None, transpose=False, use_global_buffer=False,
                   cache_absmax=True, slot=None):
    """Dense [out,in] (or [in,out] when transpose) matrix in quant_state.dtype: ONE launch for the
    nested-absmax + NF4 decode that the reference does in three (utils.py:650-675)."""
    qs = quant_state
    _lib.require_gpu(packed)
    if qs.quant_type != "nf4":
        raise NotImplementedError(f"quant_type {qs.quant_type!r}: only nf4 is implemented")
    rows, cols = qs.shape
    dtype = qs.dtype
    shape = (cols, rows) if transpose else (rows, cols)
    mirror_here = False
    if out is None and use_global_buffer and not transpose and mirror_wanted(qs):
        hit = getattr(qs, "_resident", None)
        if hit is not None and hit.dtype == dtype and hit.device == packed.device:
            return hit
        out = torch.empty(shape, dtype=dtype, device=packed.device)
        mirror_here = True
    if out is None:
        if use_global_buffer:
            if slot is None:
                slot = 1 if transpose else 0
            out = scratch(packed.device, rows * cols, dtype, slot=slot).view(shape)
        else:
            out = torch.empty(shape, dtype=dtype, device=packed.device)
    else:
        assert tuple(out.shape) == tuple(shape) and out.dtype == dtype and out.stride(1) == 1
        assert transpose or out.is_contiguous(), "a strided destination is only supported by the transposing kernel"
    ld_out = out.stride(0)
    lut = qs.code
    with _lib.device_ctx(packed):
        if qs.nested and not cache_absmax:
            if getattr(qs, "_offset_f", None) is None:
                qs._offset_f = float(qs.offset)   # one device sync per weight, then cached
            rc = _lib.lib().uamd_nf4_dequantize(
                _lib.ptr(packed), None, _lib.ptr(qs.absmax), _lib.ptr(qs.state2.code),
                _lib.ptr(qs.state2.absmax), qs._offset_f, qs.state2.blocksize, _lib.ptr(lut),
                _lib.ptr(out), rows, cols, qs.blocksize, _lib.dtype_code(dtype), int(transpose), ld_out,
                _lib.stream_of(packed))
        else:
            rc = _lib.lib().uamd_nf4_dequantize(
                _lib.ptr(packed), _lib.ptr(absmax_f32(qs)), None, None, None, 0.0, 0, _lib.ptr(lut),
                _lib.ptr(out), rows, cols, qs.blocksize, _lib.dtype_code(dtype), int(transpose), ld_out,
                _lib.stream_of(packed))
    _lib.check(rc, "uamd_nf4_dequantize")
    if mirror_here:
        qs._resident = out
        _MIRRORED.add(qs)
    return out


def dequantize_nf4_group(packed_list, qs_list, outs):
    """Row-major decode of the weights of ONE grouped GEMM (q | k | v, gate | up) into `outs` with one launch per four weights
    (uamd_nf4_dequantize_multi: the single decodes of the small ones are latency-bound). Bit-identical to dequantize_nf4 per
    weight, and falls back to it for anything the grouped launch does not take (nested absmax not cached yet on a
    non-default path, ragged sizes, 32-bit outputs). The reference decodes weight by weight (kernels/utils.py:650-675)."""
    first = qs_list[0]
    ok = len(qs_list) > 1 and all(
        q.quant_type == "nf4" and q.blocksize == first.blocksize and q.dtype == first.dtype and
        q.dtype in (torch.bfloat16, torch.float16) and (q.shape[0] * q.shape[1]) % 8192 == 0 and
        o.is_contiguous() and tuple(o.shape) == tuple(q.shape) and o.dtype == q.dtype
        for q, o in zip(qs_list, outs))
    if not ok:
        for pk, q, o in zip(packed_list, qs_list, outs):
            dequantize_nf4(pk, q, out=o)
        return outs
    _lib.require_gpu(packed_list[0])
    L = _lib.lib()
    with _lib.device_ctx(packed_list[0]):
        for i in range(0, len(qs_list), 4):
            pk, qs, os_ = packed_list[i:i + 4], qs_list[i:i + 4], outs[i:i + 4]
            n = len(qs)
            vp = _lib.ctypes.c_void_p
            a_pk = (vp * n)(*[_lib.ptr(x) for x in pk])
            a_am = (vp * n)(*[_lib.ptr(absmax_f32(q)) for q in qs])
            a_out = (vp * n)(*[_lib.ptr(o) for o in os_])
            a_n = (_lib.ctypes.c_int64 * n)(*[q.shape[0] * q.shape[1] for q in qs])
            a_lut = (vp * n)(*[_lib.ptr(q.code) for q in qs])
            rc = L.uamd_nf4_dequantize_multi(n, a_pk, a_am, a_out, a_n, a_lut, first.blocksize,
                                             _lib.dtype_code(first.dtype), _lib.stream_of(pk[0]))
            _lib.check(rc, "uamd_nf4_dequantize_multi")
    return outs


class Params4bit(torch.nn.Parameter):
    """uint8 storage + `.quant_state`, what get_lora_parameters reads (utils.py:352)."""

    def __new__(cls, data, quant_state=None):
        self = torch.Tensor._make_subclass(cls, data, False)
        self.quant_state = quant_state
        return self

    def __deepcopy__(self, memo):
        return type(self)(self.data.clone(), self.quant_state)


class Linear4bit(torch.nn.Module):
    """Frozen NF4 linear layer (bias-free or with a dense bias), bitsandbytes.nn.Linear4bit's role."""

    def __init__(self, in_features, out_features, packed, quant_state, bias=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Params4bit(packed, quant_state)
        self.bias = bias
        self.compute_dtype = quant_state.dtype

    @classmethod
    def from_linear(cls, linear, blocksize=64, compress_statistics=True):
        packed, qs = quantize_nf4(linear.weight.data, blocksize, compress_statistics)
        return cls(linear.in_features, linear.out_features, packed, qs, linear.bias)

    def forward(self, x):
        """A frozen NF4 projection on its own (LoRA on a subset of the projections -- target_modules=["q_proj", "v_proj"] --
        leaves the others as this module). Through the autograd Function: the gradient must flow THROUGH a frozen layer
        (dX = dY @ W, bitsandbytes' MatMul4Bit.backward); a bare kernel call would cut the graph and silently stop training
        everything in front of it."""
        if torch.is_grad_enabled() and x.requires_grad:
            from .kernels.fast_lora import LoRA_W
            if self.bias is not None:
                return LoRA_W.apply(x, self.weight, self.weight.quant_state, None, None, None, self.bias)
            return LoRA_W.apply(x, self.weight, self.weight.quant_state, None, None, None)
        from .kernels.utils import matmul_lora
        out = matmul_lora(x, self.weight, self.weight.quant_state, None, None, None)
        return out if self.bias is None else out + self.bias

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, nf4"


def nf4_bytes_per_param(blocksize=64, blocksize2=256):
    """0.5 + 1/64 + 4/16384 = 0.515869 B/param."""
    return 0.5 + 1.0 / blocksize + 4.0 / (blocksize * blocksize2)
