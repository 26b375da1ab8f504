"""Host-side mirror of unsloth/kernels/utils.py for the MI355X path.

Same names and argument meaning as the reference:
  QUANT_STATE                utils.py:319-320
  get_lora_parameters(_bias) utils.py:335-440   (the plug-in contract with PEFT)
  fast_dequantize            utils.py:567-679
  matmul_lora                utils.py:1128-1170
  calculate_settings         utils.py:114-133   (kept for API compatibility; the HIP kernels have no
                                                 65536-column limit, so nothing here raises on size)
plus the grouped entry points the manual-autograd Functions in fast_lora.py use
(`lora_linear_forward`, `lora_linear_dx`), which is where the MI355X design differs: projections
that share their input (q/k/v, gate/up) are ONE uamd_lora_xa launch and ONE grouped MFMA GEMM
launch, with NF4 decode and the LoRA term fused into the GEMM (csrc/gemm.hip).
"""
import ctypes
import os

import torch

from .. import _lib
from .. import nf4 as _nf4
from .._lib import GemmGroup

MAX_FUSED_SIZE = 65536
next_power_of_2 = lambda n: 1 << (max(int(n), 1) - 1).bit_length()

# NF4 forward policy. Decoding NF4 inside the GEMM (csrc/gemm.hip, no bf16 copy of W in HBM) repeats the decode
# once per 128-row M tile, so it only pays while the launch is weight-bandwidth-bound (few tokens). From
# FUSED_NF4_MAX_M tokens on, W is decoded ONCE into a per-device bf16 scratch (2.5 B/param of HBM traffic,
# ~3 % of the GEMM time at 8192 tokens) and the dense 256x256 LDS-DMA kernel runs at ~2x the fused rate.
FUSED_NF4 = True
FUSED_NF4_MAX_M = 512
# dense GEMM kernel selection: "auto" = the 256x256 ping-pong kernel (csrc/gemm256.hip: LDS-DMA, 8 waves in two
# anti-phase groups) once the launch has enough 256x256 tiles to fill the 256 CUs (GEMM256_MIN_TILES), else the
# 128x128 register-staged kernel (csrc/gemm.hip); "on"/"off" force it.
GEMM256_MODE = "auto"
# (160 since round 4: Qwen2-VL's ViT linears with N = 1280 at 4096 patches are 160 half-height tiles -- on the 128 x 128 kernel
# they ran at 0.18 PFLOP/s, the config-4 step is 2.7 % faster with them on the 256 family: profiles/r04p_config4_min_tiles_ab.txt)
GEMM256_MIN_TILES = 160
# X @ A^T / dY @ B: streaming LDS-DMA kernel (csrc/lora_side.hip) for total rank <= 64, else the first version
LORA_XA_V2 = True


def calculate_settings(n):
    """(BLOCK_SIZE, num_warps) as utils.py:114-133. Informational only on this backend."""
    BLOCK_SIZE = next_power_of_2(n)
    num_warps = 4
    if BLOCK_SIZE >= 32768:
        num_warps = 32
    elif BLOCK_SIZE >= 8192:
        num_warps = 16
    elif BLOCK_SIZE >= 2048:
        num_warps = 8
    return BLOCK_SIZE, num_warps


def QUANT_STATE(W):
    return getattr(W, "quant_state", None)


# utils.py:325-332: the dtypes for which a layer's `weight_scale(_inv)` IS its quant state. A bf16 weight that still
# carries a scale (a decompressed compressed-tensors layer) must NOT get one, or the NF4 path would read a missing absmax
# (the reference's tests/test_fast_gemv_dispatch.py:38-63 pins exactly this resolution).
_FP8_WEIGHT_DTYPES = tuple(d for d in (getattr(torch, "float8_e4m3fn", None), getattr(torch, "float8_e5m2", None))
                           if d is not None)


def _resolve_quant_state(base_layer, W):
    """utils.py:351-366 / :407-422: bitsandbytes' quant_state off the weight; for a weight that is still fp8 the layer's
    weight_scale_inv / weight_scale (block size stamped the way the reference passes it along). This backend has no fp8
    GEMM: the resolution is kept for the plugin contract, the kernels refuse such a weight loudly (_refuse_fp8)."""
    W_quant = getattr(W, "quant_state", None)
    if W_quant is None and W.dtype in _FP8_WEIGHT_DTYPES:
        W_quant = getattr(base_layer, "weight_scale_inv", None)
        if W_quant is None:
            W_quant = getattr(base_layer, "weight_scale", None)
    if getattr(base_layer, "quant_method", None) == "fp8":
        W.block_size = getattr(base_layer, "block_size", [128, 128])
        if W_quant is not None:
            W_quant.block_size = W.block_size
    return W_quant


def _refuse_fp8(projs):
    for p in projs:
        if p[0].dtype in _FP8_WEIGHT_DTYPES:
            raise NotImplementedError(
                "fp8 base weights (weight_scale / weight_scale_inv quant state) are outside this backend's scope: the MI355X "
                "path takes NF4 (bitsandbytes layout) or 16-bit base weights. The reference routes them to "
                "unsloth/kernels/fp8.py (utils.py:1146-1152).")


def get_lora_parameters(proj):
    """(W, quant_state, A, B, scaling) of a PEFT-style LoRA layer; utils.py:335-397.
    Disabled / merged adapters -> (W, quant_state, None, None, None) (:369-370)."""
    base_layer = getattr(proj, "base_layer", proj)
    W = base_layer.weight
    if hasattr(base_layer, "weight_fake_quantizer"):
        fq = getattr(base_layer, "weight_fake_quantizer", None)
        if fq is not None:
            W = fq(W)
    W_quant = _resolve_quant_state(base_layer, W)
    if getattr(proj, "disable_adapters", True) or proj.merged:
        return W, W_quant, None, None, None
    adapter = getattr(proj, "active_adapters", None)
    if adapter is None:
        adapter = getattr(proj, "active_adapter", ("default"))
    adapter = adapter[0]
    A = proj.lora_A[adapter].weight
    B = proj.lora_B[adapter].weight
    fq = getattr(proj.lora_A[adapter], "weight_fake_quantizer", None)
    if fq is not None:
        A = fq(A)
    fq = getattr(proj.lora_B[adapter], "weight_fake_quantizer", None)
    if fq is not None:
        B = fq(B)
    return W, W_quant, A, B, proj.scaling[adapter]


def get_lora_parameters_bias(proj):
    """utils.py:400-440: same plus the base layer's bias."""
    base_layer = getattr(proj, "base_layer", proj)
    W = base_layer.weight
    W_quant = _resolve_quant_state(base_layer, W)
    if getattr(proj, "disable_adapters", True) or proj.merged:
        return W, W_quant, None, None, None, base_layer.bias
    adapter = getattr(proj, "active_adapters", None)
    if adapter is None:
        adapter = getattr(proj, "active_adapter", ("default"))
    adapter = adapter[0]
    return (W, W_quant, proj.lora_A[adapter].weight, proj.lora_B[adapter].weight,
            proj.scaling[adapter], base_layer.bias)


@torch.inference_mode()
def fast_dequantize(W, quant_state=None, out=None, use_global_buffer=False):
    """utils.py:567-679. Dense [out,in] weight in quant_state.dtype; passthrough when
    quant_state is None (:578-579); returns the TRANSPOSE when handed `W.t()` of the packed
    storage, i.e. W.shape[0] == 1 (:678-679). With use_global_buffer the result is a view of a
    per-device scratch buffer that the next call overwrites (:608-632)."""
    if quant_state is None:
        return W
    _refuse_fp8([(W,)])
    is_transposed = W.shape[0] == 1
    packed = W.t() if is_transposed else W
    out = _nf4.dequantize_nf4(packed, quant_state, out=out, use_global_buffer=use_global_buffer)
    return out.t() if is_transposed else out


# ------------------------------------------------------------------------------------------------
# GEMM plumbing
def _group(B, C, N, ldb, absmax=None, xa=None, ld_xa=0, lb=None, R=0, scale=0.0, xk=None, bk=None, bias=None):
    # (for uamd_gemm_nn_256 `B` is [K, N] and `bk` is [Rk, N]; the struct is the same)
    """One uamd_gemm_group. The LoRA term comes either as (xa fp32, lb, scale) -- the register prologue of the
    128x128 kernels -- or as the rank block (xk, bk) the 256x256 kernel contracts as extra K tiles; `xa` and `R` are
    kept in both cases (the benchmark's flop accounting reads them)."""
    return GemmGroup(
        B=B.data_ptr(), C=C.data_ptr(), absmax=absmax.data_ptr() if absmax is not None else None,
        lora_xa=xa.data_ptr() if xa is not None else None, lora_b=lb.data_ptr() if lb is not None else None,
        ldb=ldb, ldc=C.stride(0), ld_xa=ld_xa, ld_lb=lb.stride(0) if lb is not None else 0,
        N=N, R=R, lora_scale=float(scale), _pad=0,
        lora_xk=xk.data_ptr() if xk is not None else None, lora_bk=bk.data_ptr() if bk is not None else None,
        ld_xk=xk.stride(0) if xk is not None else 0, ld_bk=bk.stride(0) if bk is not None else 0,
        Rk=xk.shape[1] if xk is not None else 0, _pad2=0, bias=bias.data_ptr() if bias is not None else None)


def _use_gemm256(M, K, Ns):
    """`Ns`: output widths of the groups of one launch."""
    if GEMM256_MODE == "off" or K % 64:
        return False
    if GEMM256_MODE == "on":
        return True
    cols = sum((n + 255) // 256 for n in Ns)
    # 256 x 256 tiles when they fill the chip, else the same kernel family's 128 x 256 tiles (csrc/gemm256.hip picks
    # the height); below that the 128 x 128 register-staged kernel
    return ((M + 255) // 256) * cols >= GEMM256_MIN_TILES or ((M + 127) // 128) * cols >= GEMM256_MIN_TILES


def _launch_gemm(X2d, groups, nf4, accumulate=False, nn=False):
    """`nn`: the groups' B (and rank-block BK) are [K, N] row-major, C = A @ B (uamd_gemm_nn_256); callers check
    `_use_gemm256` first -- only the 256-tile kernel family has that form."""
    arr = (GemmGroup * len(groups))(*groups)
    L = _lib.lib()
    M, K = X2d.shape
    # a LoRA term given as (fp32 XA, LB) -- no rank block -- is the register prologue of the 128 x 128 kernels only
    prologue_lora = any(g.lora_xa and not g.lora_xk for g in groups)
    if nn:
        fn, name = L.uamd_gemm_nn_256, "uamd_gemm_nn_256"
    elif nf4:
        fn, name = L.uamd_gemm_nt_nf4, "uamd_gemm_nt_nf4"
    elif _use_gemm256(M, K, [g.N for g in groups]) and not prologue_lora:
        fn, name = L.uamd_gemm_nt_256, "uamd_gemm_nt_256"
    else:
        fn, name = L.uamd_gemm_nt, "uamd_gemm_nt"
    with _lib.device_ctx(X2d):
        rc = fn(_lib.ptr(X2d), X2d.stride(0), M, K, arr, len(groups), int(accumulate),
                _lib.dtype_code(X2d.dtype), _lib.stream_of(X2d))
    _lib.check(rc, name)
    return name


def _rows2d(X):
    X2d = X.reshape(-1, X.shape[-1])
    if X2d.stride(1) != 1 or (X2d.stride(0) % 8) or (X2d.data_ptr() % 16):
        X2d = X2d.contiguous()
    return X2d


# LoRA factors are fp32 parameters used in the activation dtype (utils.py:1166-1167). A training step reads each
# factor up to 3 times (forward, checkpoint recompute, backward): convert once per parameter UPDATE instead of
# once per use. An entry is valid while (a) the parameter object, its _version and its storage are unchanged and
# (b) no optimizer has stepped since (fused optimizers update in place WITHOUT bumping _version, so a global
# optimizer post-step hook and every top-level model forward advance an epoch that invalidates everything).
import weakref
_CAST_CACHE = {}     # id(param) -> (weakref, version, data_ptr, epoch, {(tag, dtype): tensor})
_CAST_EPOCH = [0]


def invalidate_cast_cache(*_a, **_k):
    _CAST_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(invalidate_cast_cache)
except Exception:  # pragma: no cover
    pass


class _PreparedFactors:
    """Per (device, dtype): activation-dtype row-major + transposed copies of every LoRA factor seen so far,
    refreshed by ONE `uamd_lora_prepare` launch per optimizer step (epoch)."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.params = {}        # id -> [weakref, rm, tr, version, epoch, padspec]
        self.pad_bufs = {}      # key -> zero-initialised [rows, width] buffer (rank-block BK operands of the GEMM)
        self.epoch = -1
        self.table = None       # (ptr signature, descs tensor, prefix tensor, total tiles)

    @staticmethod
    def _tiles(P):
        return ((P.shape[0] + 31) // 32) * ((P.shape[1] + 31) // 32)

    _DESC = [("src", "<u8"), ("rm", "<u8"), ("tr", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("pad", "<u8"),
             ("pad_ld", "<i8"), ("pad_scale", "<f4"), ("pad_t", "<i4")]          # uamd_lora_prep_desc

    def _launch(self, ents):
        import numpy as np
        # the table only changes when a factor / copy / pad buffer moves: a cheap identity key first, the numpy
        # descriptor build (1.2 ms of host time for 448 factors -- visible at 2048 tokens per step, where the host is
        # barely ahead of the GPU) only on a miss
        key = tuple((P.data_ptr(), P.shape[0], P.shape[1], rm.data_ptr(), tr.data_ptr(),
                     None if pad is None else (pad[0].data_ptr(), pad[0].stride(0)) + tuple(pad[1:]))
                    for (P, rm, tr, pad) in ents) if len(ents) > 1 else None
        if key is not None and self.table is not None and len(self.table) > 4 and self.table[4] == key:
            tab = self.table
            any_p = ents[0][0]
            with _lib.device_ctx(any_p):
                rc = _lib.lib().uamd_lora_prepare(_lib.ptr(tab[1]), _lib.ptr(tab[2]), len(ents), tab[3],
                                                  _lib.dtype_code(self.dtype), _lib.stream_of(any_p))
            _lib.check(rc, "uamd_lora_prepare")
            return
        descs = np.zeros(len(ents), dtype=np.dtype(self._DESC))
        assert descs.dtype.itemsize == 56
        prefix, tot = np.zeros(len(ents), dtype=np.int32), 0
        for i, (P, rm, tr, pad) in enumerate(ents):
            if pad is None:
                descs[i] = (P.data_ptr(), rm.data_ptr(), tr.data_ptr(), P.shape[0], P.shape[1], 0, 0, 0.0, 0)
            else:
                buf, col, scale, transposed = pad
                descs[i] = (P.data_ptr(), rm.data_ptr(), tr.data_ptr(), P.shape[0], P.shape[1],
                            buf.data_ptr() + col * buf.element_size(), buf.stride(0), scale, int(transposed))
            prefix[i] = tot
            tot += self._tiles(P)
        sig = descs.tobytes()
        if self.table is None or self.table[0] != sig:
            d = torch.from_numpy(descs.view(np.uint8).copy()).to(self.device)
            pf = torch.from_numpy(prefix).to(self.device)
            tab = (sig, d, pf, tot, key)
            if len(ents) > 1:
                self.table = tab
        else:
            tab = self.table = self.table[:4] + (key,)
        any_p = ents[0][0]
        with _lib.device_ctx(any_p):
            rc = _lib.lib().uamd_lora_prepare(_lib.ptr(tab[1]), _lib.ptr(tab[2]), len(ents), tab[3],
                                              _lib.dtype_code(self.dtype), _lib.stream_of(any_p))
        _lib.check(rc, "uamd_lora_prepare")
        # the table tensors must outlive the launch: stream-ordered free is fine for torch's caching allocator

    def _forget(self, pid):
        self.params.pop(pid, None)
        for k in [k for k in self.pad_bufs if pid in k[0]]:
            self.pad_bufs.pop(k, None)

    def _entry(self, P, padspec=None):
        """Registers P (and, optionally, its padded/scaled third copy) and makes every copy current."""
        pid = id(P)
        ent = self.params.get(pid)
        epoch = _CAST_EPOCH[0]
        with torch.no_grad():
            if ent is None or ent[0]() is not P or ent[1].shape != P.shape:
                rm = torch.empty(P.shape, dtype=self.dtype, device=self.device)
                tr = torch.empty((P.shape[1], P.shape[0]), dtype=self.dtype, device=self.device)
                ent = [weakref.ref(P, lambda _, pid=pid: self._forget(pid)), rm, tr, -1, -1, None]
                self.params[pid] = ent
            if padspec is not None:
                old = ent[5]
                if old is None or old[0] is not padspec[0] or old[1:] != padspec[1:]:
                    ent[5] = padspec
                    ent[4] = -1                                # this entry must be (re)written
            if self.epoch != epoch:
                live = []
                for e in list(self.params.values()):
                    Q = e[0]()
                    if Q is not None:
                        live.append((Q, e[1], e[2], e[5]))
                        e[3], e[4] = Q._version, epoch
                self._launch(live)
                self.epoch = epoch
            elif ent[4] != epoch or ent[3] != P._version:       # registered (or modified in place) mid-epoch
                self._launch([(P, ent[1], ent[2], ent[5])])
                ent[3], ent[4] = P._version, epoch
        return ent

    def get(self, P, tag):
        ent = self._entry(P)
        return ent[1] if tag == "rowmajor" else ent[2]

    def get_pad(self, members, rows, width, transposed, by_rows=False):
        """members: [(P, off, scale)] written into ONE zero-initialised [rows, width] buffer: scale * P
        (transposed=False) or scale * P^T (transposed=True), side by side at COLUMN `off` (by_rows=False) or stacked
        at ROW `off` (by_rows=True). Returns the buffer: the BK operand of the GEMM's rank-block K tiles
        ([N, Rk] for the NT form, [Rk, N] for the NN form)."""
        key = (tuple(id(P) for P, _, _ in members), rows, width, bool(transposed), bool(by_rows))
        buf = self.pad_bufs.get(key)
        if buf is None:
            buf = self.pad_bufs[key] = torch.zeros((rows, width), dtype=self.dtype, device=self.device)
        for P, off, scale in members:
            self._entry(P, (buf, int(off) * (width if by_rows else 1), float(scale), bool(transposed)))
        return buf


_PREPARED = {}
LORA_PREPARE = True


def _cached_cast(P, tag, dtype, build):
    if not isinstance(P, torch.nn.Parameter):
        return build()
    if (LORA_PREPARE and tag in ("rowmajor", "T") and P.is_cuda and P.dtype == torch.float32 and P.dim() == 2
            and P.is_contiguous() and dtype in (torch.bfloat16, torch.float16)):
        key = (P.device, dtype)
        g = _PREPARED.get(key)
        if g is None:
            g = _PREPARED[key] = _PreparedFactors(P.device, dtype)
        return g.get(P, tag)
    pid = id(P)
    ent = _CAST_CACHE.get(pid)
    if (ent is None or ent[0]() is not P or ent[1] != P._version or ent[2] != P.data_ptr()
            or ent[3] != _CAST_EPOCH[0]):
        ent = (weakref.ref(P, lambda _, pid=pid: _CAST_CACHE.pop(pid, None)), P._version, P.data_ptr(),
               _CAST_EPOCH[0], {})
        _CAST_CACHE[pid] = ent
    key = (tag, dtype)
    out = ent[4].get(key)
    if out is None:
        with torch.no_grad():
            out = build()
        ent[4][key] = out
    return out


def rank_block_bk(members, rows, width, transposed, dtype, by_rows=False):
    """BK operand of the GEMM's rank-block K tiles: a zero [rows, width] buffer with scale * P (or scale * P^T when
    `transposed`) at columns col_off.. for every member (P, col_off, scale). When the factors are fp32 CUDA
    Parameters (the training case) the buffer is persistent and kept current by the once-per-step
    uamd_lora_prepare launch; otherwise it is built here."""
    prepared = LORA_PREPARE and dtype in (torch.bfloat16, torch.float16) and all(
        isinstance(P, torch.nn.Parameter) and P.is_cuda and P.dtype == torch.float32 and P.dim() == 2
        and P.is_contiguous() for P, _, _ in members)
    if not prepared:
        # plain tensors (tests, ad-hoc calls of matmul_lora): built per call with torch ops
        with torch.no_grad():
            buf = torch.zeros((rows, width), dtype=dtype, device=members[0][0].device)
            for P, off, scale in members:
                src = P.detach().t() if transposed else P.detach()
                val = (src.float() * float(scale)).to(dtype)
                if by_rows:
                    buf[off:off + src.shape[0], :src.shape[1]] = val
                else:
                    buf[:, off:off + src.shape[1]] = val
        return buf
    key = (members[0][0].device, dtype)
    g = _PREPARED.get(key)
    if g is None:
        g = _PREPARED[key] = _PreparedFactors(members[0][0].device, dtype)
    return g.get_pad(members, rows, width, transposed, by_rows)


def lora_xa(X2d, A_list, out=None, out_k=None, k_cols=0):
    """XA_g = X @ A_g^T for every projection sharing X, ONE launch. fp32, each block of columns
    padded to a multiple of 8. Returns (buffer, [(col_offset, R_padded)]). `out`: optional fp32 [M, sum Rp]
    destination with unit column stride (e.g. a column slice of a wider buffer). `out_k` (activation dtype, unit
    column stride) additionally receives the sums rounded to the activation dtype, zero-filled up to `k_cols`
    columns: the XK operand of the GEMM's rank-block K tiles."""
    dtype = X2d.dtype
    Rs = [A.shape[0] for A in A_list]
    Rp = [(r + 7) // 8 * 8 for r in Rs]
    K = X2d.shape[1]
    if len(A_list) == 1 and Rs[0] == Rp[0]:
        A0 = A_list[0]
        Acat = _cached_cast(A0, "rowmajor", dtype, lambda: A0.to(dtype).contiguous())   # A.to(dtype), utils.py:1166
    else:
        def _cat():
            if Rs == Rp:
                return torch.cat([cast_lora(A, dtype) for A in A_list], dim=0)
            buf = torch.zeros((sum(Rp), K), dtype=dtype, device=X2d.device)
            o = 0
            for A, r, rp in zip(A_list, Rs, Rp):
                buf[o:o + r] = A
                o += rp
            return buf
        if all(isinstance(A, torch.nn.Parameter) for A in A_list):
            # keyed on the first factor, tagged with the identity + version of the others
            tag = ("cat",) + tuple((id(A), A._version, A.data_ptr()) for A in A_list[1:])
            Acat = _cached_cast(A_list[0], tag, dtype, _cat)
        else:
            Acat = _cat()
    Rt = Acat.shape[0]
    if Rt > 192:
        raise NotImplementedError(f"sum of LoRA ranks sharing one input = {Rt} > 192")
    if out is None:
        out = torch.empty((X2d.shape[0], Rt), dtype=torch.float32, device=X2d.device)
    else:
        assert out.dtype == torch.float32 and tuple(out.shape) == (X2d.shape[0], Rt) and out.stride(1) == 1 \
            and out.stride(0) % 4 == 0
    L = _lib.lib()
    if out_k is not None:
        assert Rt <= 64 and out_k.dtype == dtype and out_k.stride(1) == 1 and k_cols >= Rt
        with _lib.device_ctx(X2d):
            rc = L.uamd_lora_xa2k(_lib.ptr(X2d), X2d.stride(0), _lib.ptr(Acat), Acat.stride(0), _lib.ptr(out),
                                  out.stride(0), _lib.ptr(out_k), out_k.stride(0), k_cols, X2d.shape[0], K, Rt, Rt,
                                  _lib.dtype_code(dtype), _lib.stream_of(X2d))
        _lib.check(rc, "uamd_lora_xa2k")
    else:
        fn, name = (L.uamd_lora_xa2, "uamd_lora_xa2") if (LORA_XA_V2 and Rt <= 64 and K >= 8) else (L.uamd_lora_xa, "uamd_lora_xa")
        with _lib.device_ctx(X2d):
            rc = fn(_lib.ptr(X2d), X2d.stride(0), _lib.ptr(Acat), Acat.stride(0), _lib.ptr(out), out.stride(0),
                    X2d.shape[0], K, Rt, Rt, _lib.dtype_code(dtype), _lib.stream_of(X2d))
        _lib.check(rc, name)
    offs, o = [], 0
    for rp in Rp:
        offs.append((o, rp))
        o += rp
    return out, offs


def cast_lora(P, dtype):
    """P.to(dtype) (row-major, contiguous), converted once per parameter version."""
    return _cached_cast(P, "rowmajor", dtype, lambda: P.to(dtype).contiguous())


def _pad_rank(B, Rp, dtype):
    """LoRA B [N, r] -> contiguous [N, Rp] in the activation dtype (B.to(dtype), utils.py:1167)."""
    N, r = B.shape
    if r == Rp:
        return _cached_cast(B, "rowmajor", dtype, lambda: B.to(dtype).contiguous())
    out = torch.zeros((N, Rp), dtype=dtype, device=B.device)
    out[:, :r] = B
    return out


def _rank_width(r):
    return (r + 63) // 64 * 64


def _xa_and_rank_block(X2d, A_list, want_k, out=None):
    """(xa fp32 [M, sum Rp], offsets, xk): X @ A_g^T for the projections sharing X and, when `want_k`, the same sums
    in the activation dtype zero-padded to a multiple of 64 columns (XK, the GEMM's rank-block operand)."""
    Rt = sum((A.shape[0] + 7) // 8 * 8 for A in A_list)
    if not want_k:
        xa, offs = lora_xa(X2d, A_list, out=out)
        return xa, offs, None
    M = X2d.shape[0]
    width = _rank_width(Rt)
    if Rt <= 64 and LORA_XA_V2 and X2d.shape[1] >= 8:
        xk = torch.empty((M, width), dtype=X2d.dtype, device=X2d.device)
        xa, offs = lora_xa(X2d, A_list, out=out, out_k=xk, k_cols=width)
    elif (LORA_XA_V2 and X2d.shape[1] >= 8 and Rt <= 192
          and all(A.shape[0] % 8 == 0 and A.shape[0] <= 64 for A in A_list)):
        # total rank 65 .. 192 (r = 32 on q|k|v, r = 64 on gate|up): the streaming kernel takes <= 64 rank columns per
        # launch, so the factors go in groups, each launch writing its own columns of the shared XA / rank-block buffers
        # (the last one zero-fills the padding)
        xk = torch.empty((M, width), dtype=X2d.dtype, device=X2d.device)
        xa = out if out is not None else torch.empty((M, Rt), dtype=torch.float32, device=X2d.device)
        groups, cur, cur_r = [], [], 0
        for A in A_list:
            if cur and cur_r + A.shape[0] > 64:
                groups.append((cur, cur_r))
                cur, cur_r = [], 0
            cur.append(A)
            cur_r += A.shape[0]
        groups.append((cur, cur_r))
        offs, col = [], 0
        for gi, (As, r) in enumerate(groups):
            last = gi == len(groups) - 1
            _, o = lora_xa(X2d, As, out=xa[:, col:col + r], out_k=xk[:, col:], k_cols=(width - col) if last else r)
            offs += [(col + a, b) for a, b in o]
            col += r
    else:                                   # odd ranks / very wide: first-version kernel, then a padded cast
        xa, offs = lora_xa(X2d, A_list, out=out)
        xk = torch.zeros((M, width), dtype=X2d.dtype, device=X2d.device)
        xk[:, :Rt] = xa
    return xa, offs, xk


def lora_linear_forward(X, projs, outs=None, return_xa=False, pre_xa=None):
    """Y_g = X @ W_g^T + s_g * (X @ A_g^T) @ B_g^T for projections `projs` = [(W, W_quant, A, B, s)]
    that share X. Returns a list of [.., N_g] tensors. This is matmul_lora (utils.py:1128-1170)
    for q/k/v or gate/up at once."""
    _refuse_fp8(projs)
    _lib.require_gpu(X)
    dtype = X.dtype
    if dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError("the MFMA GEMM path takes bf16/fp16 activations")
    X2d = _rows2d(X)
    M, K = X2d.shape
    lead = X.shape[:-1]
    # a projection may carry the base layer's bias as a 6th element (get_lora_parameters_bias order): it is added in the
    # GEMM epilogue, in fp32, before the single rounding (Qwen2's q/k/v; the reference leaves such layers un-fused)
    biases = [(p[5] if len(p) > 5 else None) for p in projs]
    projs = [tuple(p[:5]) for p in projs]
    with_lora = [p for p in projs if p[2] is not None]
    Ns = [(q.shape[0] if q is not None else W.shape[0]) for (W, q, _, _, _) in projs]
    fused_nf4 = [q is not None and FUSED_NF4 and M < FUSED_NF4_MAX_M and q.blocksize == 64 and K % 64 == 0
                 for (_, q, _, _, _) in projs]
    # the dense groups go to the 256x256 kernel when the launch is large enough: there the LoRA term rides along as
    # extra K tiles (XK = T(X A^T) zero-padded to 64 columns, BK_g = T(s_g B_g) at its rank columns)
    dense_Ns = [n for n, f in zip(Ns, fused_nf4) if not f]
    use256 = bool(dense_Ns) and _use_gemm256(M, K, dense_Ns)
    xa, offs, xk = None, [], None
    if with_lora and pre_xa is not None:
        # X @ A^T already produced by the kernel that produced X (glu_fwd_xa: the gated activation computes h @ A_down^T
        # while h is in its registers); the rank block must be there iff this launch wants it
        xa, offs, xk = pre_xa
        if (xk is not None) != use256:
            xa, offs, xk = _xa_and_rank_block(X2d, [p[2] for p in with_lora], use256)
    elif with_lora:
        xa, offs, xk = _xa_and_rank_block(X2d, [p[2] for p in with_lora], use256)
    results, dense_groups, nf4_groups, keep = [], [], [], []
    resident = None
    if len(projs) > 1 and all(q is not None and _nf4.mirror_wanted(q) for (_, q, _, _, _) in projs) and not any(fused_nf4):
        _, resident = _nf4.resident_group([p_[0] for p_ in projs], [p_[1] for p_ in projs])
    # the members that are decoded for this launch: ONE decode launch for all of them (nf4.dequantize_nf4_group), each into the
    # scratch slot of its own that the grouped GEMM reads
    decoded = {}
    if resident is None:
        todo = [gi for gi, (W, q, _, _, _) in enumerate(projs) if q is not None and not fused_nf4[gi]]
        if len(todo) > 1 and all(projs[gi][1].dtype == dtype and projs[gi][1].quant_type == "nf4" for gi in todo):
            slots = [_nf4.scratch(X.device, projs[gi][1].shape[0] * projs[gi][1].shape[1], dtype, slot=8 + gi).view(
                tuple(projs[gi][1].shape)) for gi in todo]
            _nf4.dequantize_nf4_group([projs[gi][0] for gi in todo], [projs[gi][1] for gi in todo], slots)
            decoded = dict(zip(todo, slots))
    li = 0
    for gi, (W, W_quant, A, B, s) in enumerate(projs):
        if W_quant is not None:
            assert W_quant.shape[1] == K, "weight/in_features mismatch"
        N = Ns[gi]
        C = outs[gi] if outs is not None else torch.empty((M, N), dtype=dtype, device=X.device)
        kw = {}
        if biases[gi] is not None:
            bias = biases[gi].detach()
            if bias.dtype != dtype or not bias.is_contiguous():
                bias = bias.to(dtype).contiguous()
            assert bias.numel() == N
            keep.append(bias)
            kw["bias"] = bias
        if A is not None:
            o, rp = offs[li]
            li += 1
            if xk is not None and not fused_nf4[gi]:
                bk = rank_block_bk([(B, o, s)], N, xk.shape[1], False, dtype)
                keep.append(bk)
                kw.update(xa=xa[:, o:], ld_xa=xa.stride(0), R=rp, scale=s, xk=xk, bk=bk)
            else:
                lb = _pad_rank(B, rp, dtype)
                keep.append(lb)
                kw.update(xa=xa[:, o:], ld_xa=xa.stride(0), lb=lb, R=rp, scale=s)
        if fused_nf4[gi]:
            nf4_groups.append(_group(W, C, N, 0, absmax=_nf4.absmax_f32(W_quant), **kw))
        else:
            Wd = W
            if resident is not None:
                Wd = resident[gi]
            elif gi in decoded:
                Wd = decoded[gi]
            elif W_quant is not None:
                # one scratch slot per group member: the grouped launch reads all of them
                Wd = _nf4.dequantize_nf4(W, W_quant, use_global_buffer=True, slot=8 + gi)
                if Wd.dtype != dtype:
                    raise TypeError(f"quant_state.dtype {Wd.dtype} != activation dtype {dtype}: the dequantised "
                                    "weight would be misread by the GEMM (set quant_state.dtype to the compute dtype)")
            elif Wd.dtype != dtype or Wd.stride(1) != 1 or Wd.stride(0) % 8:
                Wd = Wd.to(dtype).contiguous()
            keep.append(Wd)
            dense_groups.append(_group(Wd, C, N, Wd.stride(0), **kw))
        results.append(C)
    if nf4_groups:
        _launch_gemm(X2d, nf4_groups, nf4=True)
    if dense_groups:
        _launch_gemm(X2d, dense_groups, nf4=False)
    results = [r.view(*lead, r.shape[-1]) for r in results]
    if return_xa:
        # per projection: fp32 [M, Rp] view of X @ A^T (None without adapter), kept for d_B = s * dY^T (X A^T)
        views, li = [], 0
        for (W, W_quant, A, B, s) in projs:
            if A is None:
                views.append(None)
            else:
                o, rp = offs[li]
                li += 1
                views.append(xa[:, o:o + rp])
        return results, views
    return results


def lora_dx_terms(dYs, projs):
    """P_g = dY_g @ B_g (fp32 [M, Rp]) for every projection with an adapter (None otherwise): the rank-r factor
    shared by dX += s (dY B) A and by d_A = s (dY B)^T X (fast_lora.py:172-189).
    When the dX GEMM will run on the 256x256 kernel, each launch also writes T(P_g) into a shared zero-padded
    [M, 64k] rank block (attached to the returned tensor as `_uamd_xk = (xk, col_off)`): the XK operand of that
    GEMM's extra K tiles."""
    terms = []
    projs = [tuple(p[:5]) for p in projs]              # (a 6th element, the bias, plays no part in any gradient but its own)
    ranks = [None if A is None else A.shape[0] for (_, _, A, _, _) in projs]
    dY2 = [_rows2d(dY) for dY in dYs]
    M = dY2[0].shape[0]
    dev, dtype = dY2[0].device, dY2[0].dtype
    Kin = [(q.shape[1] if q is not None else W.shape[1]) for (W, q, _, _, _) in projs]
    want_k = (dtype in (torch.bfloat16, torch.float16) and LORA_XA_V2
              and all(_use_gemm256(M, 64, [k]) for k in Kin))
    # all P_g side by side in ONE [M, sum r] buffer when possible: lora_linear_dx can then hand the whole rank
    # block to a single K-concatenated GEMM
    shared, col = None, 0
    with_lora = [r for r in ranks if r is not None]
    if len(projs) > 1 and len(with_lora) == len(projs) and all(r % 8 == 0 for r in ranks):
        shared = torch.empty((M, sum(ranks)), dtype=torch.float32, device=dev)
    Rt = sum((r + 7) // 8 * 8 for r in with_lora)
    xk = None
    if want_k and with_lora and Rt <= 192 and max(with_lora) <= 64 and (shared is not None or len(with_lora) == 1):
        xk = torch.empty((M, _rank_width(Rt)), dtype=dtype, device=dev)
    n_left = len(with_lora)
    for dY2d, (W, W_quant, A, B, s) in zip(dY2, projs):
        if A is None:
            terms.append(None)
            continue
        n_left -= 1
        Bt = _cached_cast(B, "T", dtype, lambda B=B, dtype=dtype: B.to(dtype).t().contiguous())        # [r, N]
        r = A.shape[0]
        rp = (r + 7) // 8 * 8
        kw = {}
        if xk is not None:
            # the last launch also zero-fills the padding columns up to the 64-multiple
            kw = dict(out_k=xk[:, col:], k_cols=(xk.shape[1] - col) if n_left == 0 else rp)
        if shared is not None:
            xa, offs = lora_xa(dY2d, [Bt], out=shared[:, col:col + r], **kw)
        else:
            xa, offs = lora_xa(dY2d, [Bt], **kw)                    # dY @ B
        t = xa[:, :offs[0][1]]
        if xk is not None:
            t._uamd_xk = (xk, col)
        elif want_k:                            # odd ranks / total rank > 64: padded cast with torch ops
            own = torch.zeros((M, _rank_width(t.shape[1])), dtype=dtype, device=dev)
            own[:, :t.shape[1]] = t
            t._uamd_xk = (own, 0)
        col += rp
        terms.append(t)
    return terms


MERGE_DX = True
# dX = dY @ W through the NN form of the 256-tile GEMM (row-major decode, transposing LDS reads) instead of a
# transposed decode + the NT form
NN_DX = True


def _adjacent_columns(ts):
    """The 2-D views `ts` are consecutive column blocks of one row-major buffer -> the [M, sum N] view, else None."""
    t0 = ts[0]
    off = 0
    for t in ts:
        if (t.dim() != 2 or t.stride(1) != 1 or t.stride(0) != t0.stride(0) or t.shape[0] != t0.shape[0]
                or t.dtype != t0.dtype or t.device != t0.device
                or t.data_ptr() != t0.data_ptr() + off * t0.element_size()):
            return None
        off += t.shape[1]
    if off > t0.stride(0):
        return None
    return torch.as_strided(t0, (t0.shape[0], off), (t0.stride(0), 1))


def _lora_linear_dx_merged(dYs, projs, out, terms):
    """dX = [dY_1 | dY_2 | ...] @ [W_1; W_2; ...] + s [P_1 | P_2 | ...] @ [A_1; A_2; ...] as ONE GEMM when the
    incoming gradients are column blocks of one buffer (the attention backward writes dQ, dK, dV that way): the
    contraction simply runs over the concatenated output features. Saves the read-modify-write of dX per extra
    projection and the short-K launches (k_proj / v_proj: K = 1024). Returns None when the shapes do not allow it."""
    if len(dYs) < 2 or any(q is None for (_, q, _, _, _) in projs):
        return None
    if any(A is None for (_, _, A, _, _) in projs) or any(t is None for t in terms):
        return None
    dY2 = [_rows2d(dY) for dY in dYs]
    dYcat = _adjacent_columns(dY2)
    Pcat = _adjacent_columns(list(terms))
    if dYcat is None or Pcat is None or dYcat.shape[1] % 64 or dYcat.stride(0) % 8:
        return None
    dtype = dYcat.dtype
    Kin = projs[0][1].shape[1]
    if any(q.shape[1] != Kin or q.dtype != dtype for (_, q, _, _, _) in projs):
        return None
    M, Ntot = dYcat.shape
    A_list = [A for (_, _, A, _, _) in projs]
    if out is None:
        out = torch.empty((M, Kin), dtype=dtype, device=dYcat.device)
    xks = [getattr(t, "_uamd_xk", None) for t in terms]
    have_xk = all(x is not None and x[0] is xks[0][0] for x in xks)
    if NN_DX and have_xk and _use_gemm256(M, Ntot, [Kin]) and Kin % 8 == 0:
        # NN form: [W_q; W_k; W_v] stacked by ROWS is just the three row-major decodes one after the other -- the
        # layout the forward uses -- and the GEMM contracts over those rows (no transposed copy of any weight)
        if all(_nf4.mirror_wanted(p_[1]) for p_ in projs):
            Wcat, _ = _nf4.resident_group([p_[0] for p_ in projs], [p_[1] for p_ in projs])
        else:
            Wcat = _nf4.scratch(dYcat.device, Ntot * Kin, dtype, slot=2).view(Ntot, Kin)
            row, rows_of = 0, []
            for (W, q, _, _, _) in projs:
                rows_of.append(Wcat[row:row + q.shape[0]])
                row += q.shape[0]
            _nf4.dequantize_nf4_group([p_[0] for p_ in projs], [p_[1] for p_ in projs], rows_of)
        xk = xks[0][0]
        bk = rank_block_bk([(A, x[1], s) for (_, _, A, _, s), x in zip(projs, xks)], xk.shape[1], Kin, False, dtype,
                           by_rows=True)                      # [s_q A_q; s_k A_k; s_v A_v; 0] : [Rk, Kin]
        g = _group(Wcat, out, Kin, Wcat.stride(0), xa=Pcat, ld_xa=Pcat.stride(0), R=Pcat.shape[1], scale=1.0, xk=xk, bk=bk)
        _launch_gemm(dYcat, [g], nf4=False, accumulate=False, nn=True)
        return out
    Wt = _nf4.scratch(dYcat.device, Kin * Ntot, dtype, slot=2).view(Kin, Ntot)
    col = 0
    for (W, q, _, _, _) in projs:
        n = q.shape[0]
        _nf4.dequantize_nf4(W, q, out=Wt[:, col:col + n], transpose=True)          # [Kin, n] at column `col`
        col += n
    if _use_gemm256(M, Ntot, [Kin]) and have_xk:
        # rank block as extra K tiles: XK = [T(P_q) | T(P_k) | T(P_v) | 0], BK = [s_q A_q^T | s_k A_k^T | s_v A_v^T | 0]
        xk = xks[0][0]
        bk = rank_block_bk([(A, x[1], s) for (_, _, A, _, s), x in zip(projs, xks)], Kin, xk.shape[1], True, dtype)
        g = _group(Wt, out, Kin, Wt.stride(0), xa=Pcat, ld_xa=Pcat.stride(0), R=Pcat.shape[1], scale=1.0, xk=xk, bk=bk)
    else:
        scales = {float(s) for (_, _, _, _, s) in projs}
        if len(scales) != 1:
            return None
        tag = ("catT",) + tuple((id(A), A._version, A.data_ptr()) for A in A_list[1:])
        lb = _cached_cast(A_list[0], tag, dtype,
                          lambda: torch.cat([_cached_cast(A, "T", dtype, lambda A=A: A.to(dtype).t().contiguous())
                                             for A in A_list], dim=1).contiguous())
        g = _group(Wt, out, Kin, Wt.stride(0), xa=Pcat, ld_xa=Pcat.stride(0), lb=lb, R=Pcat.shape[1],
                   scale=scales.pop())
    _launch_gemm(dYcat, [g], nf4=False, accumulate=False)
    return out


def lora_linear_dx(dYs, projs, out=None, terms=None):
    """dX = sum_g dY_g @ W_g + s_g * (dY_g @ B_g) @ A_g   (fast_lora.py:193-204, 497-517, 639-647).
    The contraction runs over `out`, so the NF4 weight is decoded TRANSPOSED into the per-device
    scratch (one launch) and fed to the same NT GEMM. `out` (e.g. the saved X, reference's
    inplace=True) receives the result. `terms` = lora_dx_terms(dYs, projs) when the caller already has them."""
    dtype = dYs[0].dtype
    projs = [tuple(p[:5]) for p in projs]
    _refuse_fp8(projs)
    if terms is None:
        terms = lora_dx_terms(dYs, projs)
    merged = _lora_linear_dx_merged(dYs, projs, out, terms) if MERGE_DX else None
    if merged is not None:
        return merged
    first = True
    for dY, (W, W_quant, A, B, s), xa in zip(dYs, projs, terms):
        dY2d = _rows2d(dY)
        M, N = dY2d.shape
        Kin = W_quant.shape[1] if W_quant is not None else W.shape[1]
        xk = getattr(xa, "_uamd_xk", None) if A is not None else None
        if NN_DX and _use_gemm256(M, N, [Kin]) and Kin % 8 == 0 and (A is None or xk is not None):
            # NN form: contract over the rows of the [N, Kin] weight as the forward decodes it
            if W_quant is not None:
                Wd = _nf4.dequantize_nf4(W, W_quant, use_global_buffer=True)
            else:
                Wd = W if (W.dtype == dtype and W.stride(1) == 1 and W.stride(0) % 8 == 0) else W.to(dtype).contiguous()
            if Wd.dtype != dtype:
                raise TypeError(f"quant_state.dtype {Wd.dtype} != activation dtype {dtype}: the dequantised weight "
                                "would be misread by the GEMM (set quant_state.dtype to the compute dtype)")
            if out is None:
                out = torch.empty((M, Kin), dtype=dtype, device=dY.device)
            kw = {}
            if A is not None:
                bk = rank_block_bk([(A, xk[1], s)], xk[0].shape[1], Kin, False, dtype, by_rows=True)   # s A at its rank rows
                kw = dict(xa=xa, ld_xa=xa.stride(0), R=xa.shape[1], scale=s, xk=xk[0], bk=bk)
            _launch_gemm(dY2d, [_group(Wd, out, Kin, Wd.stride(0), **kw)], nf4=False, accumulate=not first, nn=True)
            first = False
            continue
        if W_quant is not None:
            Wt = _nf4.dequantize_nf4(W, W_quant, transpose=True, use_global_buffer=True)   # [Kin, N]
        else:
            Wt = W.to(dtype).t().contiguous()
        if out is None:
            out = torch.empty((M, Kin), dtype=dtype, device=dY.device)
        if Wt.dtype != dtype:
            raise TypeError(f"quant_state.dtype {Wt.dtype} != activation dtype {dtype}: the dequantised weight "
                            "would be misread by the GEMM (set quant_state.dtype to the compute dtype)")
        kw = {}
        if A is not None:
            rp = xa.shape[1]
            xk = getattr(xa, "_uamd_xk", None)
            if xk is not None and _use_gemm256(M, N, [Kin]):
                bk = rank_block_bk([(A, xk[1], s)], Kin, xk[0].shape[1], True, dtype)     # s A^T at its rank columns
                kw = dict(xa=xa, ld_xa=xa.stride(0), R=rp, scale=s, xk=xk[0], bk=bk)
            else:
                if A.shape[0] == rp:
                    lb = _cached_cast(A, "T", dtype, lambda A=A, dtype=dtype: A.to(dtype).t().contiguous())    # A^T [Kin, r]
                else:
                    lb = _pad_rank(A.to(dtype).t(), rp, dtype)
                kw = dict(xa=xa, ld_xa=xa.stride(0), lb=lb, R=rp, scale=s)
        g = _group(Wt, out, Kin, Wt.stride(0), **kw)
        _launch_gemm(dY2d, [g], nf4=False, accumulate=not first)
        first = False
    return out


# ------------------------------------------------------------------------------------------------
# LoRA gradient products  G = s * P^T @ Z  (csrc/lora_side.hip)
def lora_tn_supported(Zs):
    return all(Z.is_cuda and Z.dtype in (torch.bfloat16, torch.float16) and Z.shape[-1] % 8 == 0 for Z in Zs)


# Gradient sinks: id(parameter) -> object with .grad_view(p) (contiguous fp32 tensor of p's shape that ACCUMULATES
# the gradient, e.g. a slice of dp.LoRAGradArena) and .ready(p) (called once the kernel that adds into it has been
# enqueued). With a sink the fused LoRA-gradient kernel adds straight into the arena: no per-parameter
# AccumulateGrad kernel, no temporary.
GRAD_SINKS = {}


def grad_sink(param):
    """The live sink that owns `param`'s gradient, or None. Entries hold WEAK references to their arena: an arena that
    is alive keeps its parameters alive, so a hit can never belong to a dead parameter whose id() was recycled."""
    if not GRAD_SINKS:
        return None
    ref = GRAD_SINKS.get(id(param))
    if ref is None:
        return None
    sink = ref()
    if sink is None:
        GRAD_SINKS.pop(id(param), None)
    return sink


def lora_tn(problems, targets=None):
    """problems: [(P fp32 [M, >=R] with unit column stride, Z [M, N], R, out_nr, scale)].
    Returns the fp32 products, [R, N] (out_nr False: lora_A.grad layout) or [N, R] (True: lora_B.grad layout).
    `targets[i]` (optional): contiguous fp32 tensor of that shape to ACCUMULATE into instead of allocating.
    One launch per 8 sixteen-rank problems; deterministic."""
    if not problems:
        return []
    M = problems[0][1].shape[0]
    dev = problems[0][1].device
    dtype = problems[0][1].dtype
    outs, descs, keep = [], [], []
    for pi, (P, Z, R, out_nr, scale) in enumerate(problems):
        Z2 = _rows2d(Z)
        assert Z2.shape[0] == M and P.shape[0] == M and P.dtype == torch.float32 and P.stride(1) == 1
        N = Z2.shape[1]
        tgt = targets[pi] if targets is not None else None
        if tgt is not None:
            assert (tgt.dtype == torch.float32 and tgt.is_contiguous() and tgt.device == dev
                    and tuple(tgt.shape) == ((N, R) if out_nr else (R, N)))
            out = tgt
        else:
            out = torch.empty((N, R) if out_nr else (R, N), dtype=torch.float32, device=dev)
        outs.append(out)
        keep.append(Z2)
        for r0 in range(0, R, 16):
            rc = min(16, R - r0)
            optr = out.data_ptr() + (4 * r0 if out_nr else 4 * r0 * N)
            descs.append(_lib.LoraTnProblem(P=P.data_ptr() + 4 * r0, Z=Z2.data_ptr(), out=optr, ldp=P.stride(0),
                                            ldz=Z2.stride(0), ldo=(R if out_nr else N), N=N, R=rc,
                                            out_nr=int(bool(out_nr)) | (2 if tgt is not None else 0),
                                            scale=float(scale)))
    S = (M + 127) // 128            # upper bound on the row chunks the kernel may choose
    L = _lib.lib()
    for i in range(0, len(descs), 8):
        chunk = descs[i:i + 8]
        need = sum(S * 16 * ((d.N + 127) // 128) * 128 for d in chunk)
        ws = _nf4.scratch(dev, need, torch.float32, slot=40)
        arr = (_lib.LoraTnProblem * len(chunk))(*chunk)
        with _lib.device_ctx(ws):
            rc = L.uamd_lora_tn(arr, len(chunk), M, _lib.ptr(ws), need, _lib.dtype_code(dtype), _lib.stream_of(ws))
        _lib.check(rc, "uamd_lora_tn")
    return outs


# the gated activation fused with the LoRA skinny products (csrc/glu.hip glu_xa_kernel). Measured at Llama-3-8B MLP widths,
# 8192 tokens, in the step (profiles/r03ab_bench_kernel_stats.csv, after the tile loop lost its per-iteration vmcnt drain):
# backward 281 us against 217 + 2 x ~65 us for the separate launches, forward 157 us against 108 + 57 us; whole step +0.3-0.5 %
# with the forward fused too (profiles/r03ab_bench_ab.txt). At 2048 tokens 128 blocks left half the chip idle (118 vs 109 us,
# profiles/r03j_glu_fused_bench.jsonl) and rounds 3-4 fused from 4096 tokens on; with the columns of a row group split over
# adjacent workgroups (round 5, csrc/glu.hip launch_xa) the fused kernels win there too -- 38 vs 55 us forward, 79 vs 109 us
# backward at 2048 tokens, 139 vs 153 / 252 vs 325 us at 8192; batch-1 step -1.7 % (profiles/r05_glu_fused_bench.txt) -- so both
# directions are fused from 2048 tokens on.
# GLU_FUSED (module attribute; tools/glu_*.py set "all"): "both" | "bwd" (backward only, round 3's first default) | "all" (both, any
# size) | False.
GLU_FUSED = "both"
GLU_FUSED_MIN_ROWS = 2048
_GLU_ACTS = {"swiglu": 0, "geglu_exact": 1, "geglu_approx": 2}


# Rows of a whole number of 4 KiB pages (14336 or 28672 bf16 columns: the gated-MLP intermediates) read as the A operand of a
# GEMM -- 256 rows x 128 bytes per K tile -- put every line of a tile on the same few memory channels: the same GEMM with 128
# bytes of row padding measured +4.3 % (NT, K = 14336), +2.7 % (NN), tools/gemm_pad_probe.py / profiles/r06_gemm_row_padding.jsonl.
# alloc_rows gives such buffers a row stride of N + ROW_PAD elements (ROW_PAD = 0: plain).
ROW_PAD = 64


def alloc_rows(M, N, dtype, device, ld=None):
    """A [M, N] row-major buffer; `ld` = forced row stride (elements), default N (+ ROW_PAD for page-multiple rows >= 16 KiB)."""
    if ld is None:
        item = torch.empty((), dtype=dtype).element_size()
        ld = N + ROW_PAD if (ROW_PAD and N * item >= 16384 and (N * item) % 4096 == 0) else N
    buf = torch.empty((M, ld), dtype=dtype, device=device)
    return buf if ld == N else buf[:, :N]


def _same_rows(ts):
    """2-D row-major views with ONE row stride (a multiple of 8 elements) and 16-byte aligned starts: what the gated-activation
    kernels take through their single `ld`"""
    t0 = ts[0]
    return all(t.dim() == 2 and t.stride(1) == 1 and t.stride(0) == t0.stride(0) and t.stride(0) % 8 == 0
               and t.data_ptr() % 16 == 0 and t.shape == t0.shape for t in ts)


def _glu_fusable(dtype, tensors, ranks, backward):
    K = tensors[0].shape[-1]
    mode = GLU_FUSED if GLU_FUSED is not True else "all"
    if not mode or (mode == "bwd" and not backward) or (mode in ("bwd", "both") and tensors[0].shape[0] < GLU_FUSED_MIN_ROWS):
        return False
    return (LORA_XA_V2 and dtype in (torch.bfloat16, torch.float16) and K % 8 == 0
            and all(t.is_cuda and t.dtype == dtype for t in tensors) and _same_rows(tensors)
            and all(r is not None and r % 8 == 0 and 0 < r <= 64 for r in ranks))


_GLU_WS = {}        # (device index, stream) -> (fp32 partials, int32 arrival counters): the column-split kernels' workspace


def _glu_workspace(t, n_products, max_rank):
    """Per device and stream (two launches in flight on different streams must not share it), grown on demand. The counters
    are zero when a launch starts and the kernel leaves them zero (the last workgroup of a row group resets its counter)."""
    M, K = t.shape
    need = int(_lib.lib().uamd_glu_xa_workspace(M, K, n_products, max_rank))
    groups = (M + 15) // 16
    key = (t.device.index, int(torch.cuda.current_stream(t.device).cuda_stream))
    ent = _GLU_WS.get(key)
    if ent is None or ent[0].numel() < need or ent[1].numel() < groups:
        part = torch.empty(max(need, ent[0].numel() if ent else 0), dtype=torch.float32, device=t.device)
        counters = torch.zeros(max(groups, ent[1].numel() if ent else 0), dtype=torch.int32, device=t.device)
        ent = _GLU_WS[key] = (part, counters)
    return ent


def glu_fwd_xa(act, e, g, down, n_out_cols_hint=None):
    """h = act(e) * g AND the down projection's X A^T (fp32 [M, r]) + rank block, in ONE pass over e and g
    (uamd_glu_fwd_xa). `down` = (W, W_quant, A, B, s[, bias]). Returns (h, pre_xa) with pre_xa = (xa, offs, xk) as
    lora_linear_forward(pre_xa=) takes it, or None when the shapes do not allow the fusion (the caller then runs the plain
    activation kernel and lets lora_linear_forward compute X A^T itself)."""
    A = down[2]
    # SwiGLU only (the hot path's activation): its fused arithmetic is bit-identical to the plain kernel, which is pinned
    # bit for bit to the reference's Triton kernel; the GeGLU instantiations differ from theirs in the last place in fp16
    if act != "swiglu" or A is None or not _glu_fusable(e.dtype, [e, g], [A.shape[0]], False):
        return None
    M, K = e.shape
    dtype = e.dtype
    r = A.shape[0]
    W, q = down[0], down[1]
    N = q.shape[0] if q is not None else W.shape[0]
    fused_nf4 = q is not None and FUSED_NF4 and M < FUSED_NF4_MAX_M and q.blocksize == 64 and K % 64 == 0
    want_k = (not fused_nf4) and _use_gemm256(M, K, [N])
    Ac = _cached_cast(A, "rowmajor", dtype, lambda: A.to(dtype).contiguous())
    h = alloc_rows(M, K, dtype, e.device, ld=e.stride(0))          # (one `ld` for e, g and h)
    xa = torch.empty((M, r), dtype=torch.float32, device=e.device)
    xk = torch.empty((M, _rank_width(r)), dtype=dtype, device=e.device) if want_k else None
    with _lib.device_ctx(e):
        part, counters = _glu_workspace(e, 1, r)
        rc = _lib.lib().uamd_glu_fwd_xa_ws(_GLU_ACTS[act], _lib.ptr(e), _lib.ptr(g), _lib.ptr(h), M, K, e.stride(0), _lib.ptr(Ac),
                                           Ac.stride(0), r, _lib.ptr(xa), xa.stride(0), r,
                                           _lib.ptr(xk) if xk is not None else None, xk.stride(0) if xk is not None else 0,
                                           xk.shape[1] if xk is not None else 0, _lib.ptr(part), part.numel(), _lib.ptr(counters),
                                           _lib.dtype_code(dtype), _lib.stream_of(e))
    if rc != 0:
        counters.zero_()                 # a launch that did not complete may leave arrival counts behind (include/unsloth_amd.h)
    _lib.check(rc, "uamd_glu_fwd_xa_ws")
    return h, (xa, [(0, r)], xk)


def glu_bwd_terms(act, DW, e, g, up, gate):
    """The in-place activation backward (DW <- h, e <- df, g <- de) AND P_up = df @ B_up, P_gate = de @ B_gate -- the
    lora_dx_terms([df, de], [up, gate]) of fast_lora.mlp_backward -- in ONE pass (uamd_glu_bwd_xa). Returns
    (h, df, de, [p_up, p_gate]) or None when the shapes do not allow the fusion."""
    (Au, Bu), (Ag, Bg) = (up[2], up[3]), (gate[2], gate[3])
    if act != "swiglu" or Au is None or Ag is None or not _glu_fusable(e.dtype, [DW, e, g], [Au.shape[0], Ag.shape[0]], True):
        return None
    M, K = e.shape
    dtype = e.dtype
    ru, rg = Au.shape[0], Ag.shape[0]
    Kin = [(p[1].shape[1] if p[1] is not None else p[0].shape[1]) for p in (up, gate)]
    want_k = all(_use_gemm256(M, 64, [k]) for k in Kin)
    But = _cached_cast(Bu, "T", dtype, lambda: Bu.to(dtype).t().contiguous())        # [r, K]
    Bgt = _cached_cast(Bg, "T", dtype, lambda: Bg.to(dtype).t().contiguous())
    shared = torch.empty((M, ru + rg), dtype=torch.float32, device=e.device)
    xk = torch.empty((M, _rank_width(ru + rg)), dtype=dtype, device=e.device) if want_k else None
    pu, pg = shared[:, :ru], shared[:, ru:]
    null = None
    with _lib.device_ctx(e):
        part, counters = _glu_workspace(e, 2, max(ru, rg))
        rc = _lib.lib().uamd_glu_bwd_xa_ws(
            _GLU_ACTS[act], _lib.ptr(DW), _lib.ptr(e), _lib.ptr(g), M, K, e.stride(0),
            _lib.ptr(But), But.stride(0), ru, _lib.ptr(pu), shared.stride(0), ru,
            _lib.ptr(xk) if want_k else null, xk.stride(0) if want_k else 0, ru if want_k else 0,
            _lib.ptr(Bgt), Bgt.stride(0), rg, _lib.ptr(pg), shared.stride(0), rg,
            _lib.ptr(xk[:, ru:]) if want_k else null, xk.stride(0) if want_k else 0, (xk.shape[1] - ru) if want_k else 0,
            _lib.ptr(part), part.numel(), _lib.ptr(counters), _lib.dtype_code(dtype), _lib.stream_of(e))
    if rc != 0:
        counters.zero_()
    _lib.check(rc, "uamd_glu_bwd_xa_ws")
    if want_k:
        pu._uamd_xk = (xk, 0)
        pg._uamd_xk = (xk, ru)
    return DW, e, g, [pu, pg]


def dense_dw(dY, X, out=None, accumulate=False):
    """dW[out, in] (+)= dY[T, out]^T @ X[T, in]: the weight gradient of a trainable dense projection (full fine-tuning;
    torch.nn.Linear's `grad_output.t().mm(input)`). `dY` may be several projections' gradients side by side in one buffer
    ([T, sum out_g]: dQ | dK | dV as the attention backward writes them) -- then `out` is their stacked gradient
    [sum out_g, in]. One uamd_gemm_tn_256 launch, both operands read in place; a token count that is not a multiple of
    64 is zero-padded (zero rows add nothing)."""
    dY2d, X2d = _rows2d(dY), _rows2d(X)
    _lib.require_gpu(X2d)
    T, N_out = dY2d.shape
    assert X2d.shape[0] == T and X2d.dtype == dY2d.dtype
    dtype = X2d.dtype
    if dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError("the MFMA GEMM path takes bf16/fp16 activations")
    N_in = X2d.shape[1]
    if N_out % 8 or N_in % 8:
        raise NotImplementedError(f"dense_dw: out {N_out} / in {N_in} features must be multiples of 8")
    if T % 64:
        pad = 64 - T % 64
        dY2d = torch.nn.functional.pad(dY2d, (0, 0, 0, pad))
        X2d = torch.nn.functional.pad(X2d, (0, 0, 0, pad))
        T += pad
    if dY2d.stride(1) != 1 or dY2d.stride(0) % 8 or dY2d.data_ptr() % 16:
        dY2d = dY2d.contiguous()
    if X2d.stride(1) != 1 or X2d.stride(0) % 8 or X2d.data_ptr() % 16:
        X2d = X2d.contiguous()
    if out is None:
        out = torch.empty((N_out, N_in), dtype=dtype, device=X2d.device)
        accumulate = False
    assert out.dtype == dtype and tuple(out.shape) == (N_out, N_in) and out.stride(1) == 1
    g = _group(X2d, out, N_in, X2d.stride(0))
    g.ldc = out.stride(0)
    arr = (GemmGroup * 1)(g)
    with _lib.device_ctx(X2d):
        rc = _lib.lib().uamd_gemm_tn_256(_lib.ptr(dY2d), dY2d.stride(0), N_out, T, arr, 1, int(bool(accumulate)),
                                        _lib.dtype_code(dtype), _lib.stream_of(X2d))
    _lib.check(rc, "uamd_gemm_tn_256")
    return out


def matmul_lora(X, W, W_quant, A, B, s, out=None):
    """utils.py:1128-1170: out = X @ dequant(W).T (+ s * (X @ A.T) @ B.T).
    `W` may be the transposed packed storage (`W.t()`, shape [1, n/2]) or a transposed dense view,
    as LoRA_MLP.backward passes it (fast_lora.py:156); then the product is X @ W and A/B arrive
    already swapped and transposed ([in? r] layout), exactly like the reference."""
    reshape = X.dim() == 3
    if reshape:
        batch, seq_len, _ = X.shape
    transposed = (W_quant is not None and W.shape[0] == 1) or (
        W_quant is None and W.dim() == 2 and W.stride(0) == 1 and W.stride(1) != 1)
    if not transposed:
        res = lora_linear_forward(X, [(W, W_quant, A, B, s)], outs=None if out is None else [out])[0]
    else:
        Wn = W.t()
        # reference call: matmul_lora(dY, W.t(), q, B.t(), A.t(), s) -> our (A, B) are (B_lora^T, A_lora^T)
        proj = (Wn, W_quant, None if B is None else B.t(), None if A is None else A.t(), s)
        res = lora_linear_dx([X], [proj], out=out)
    return res.view(batch, seq_len, -1) if reshape else res


def fast_linear_forward(proj, X, temp_lora=None, out=None):
    """utils.py:1082-1125, the decode-time linear. X [bsz, q_len, in]: a single token of a single sequence goes through
    the GEMV kernels (csrc/decode.hip: NF4 decoded in the kernel, LoRA and bias folded into its reduction -- the
    reference's cdequantize_blockwise_fp32 + cgemm_4bit_inference_naive + mv + addmv); everything else through
    matmul_lora like the reference's q_len != 1 / bsz > 1 branches. `temp_lora` is accepted and unused (the A x
    products of a launch land in one fp32 vector of their own)."""
    from . import decode as _dk
    W, W_quant, A, B, s, bias = get_lora_parameters_bias(proj)
    _refuse_fp8([(W,)])
    bsz, q_len, in_dim = X.shape
    n_out = int(W_quant.shape[0]) if W_quant is not None else int(W.shape[0])
    if bsz == 1 and q_len == 1 and in_dim <= 16384 and X.dtype in (torch.bfloat16, torch.float16) \
            and (W_quant is not None or W.dtype == X.dtype):
        flat = out.view(-1) if out is not None else None
        (y,) = _dk.linear_group(X.reshape(-1), [(W, W_quant, A, B, s, bias)], out=flat)
        return y.view(1, 1, n_out)
    y = matmul_lora(X, W, W_quant, A, B, s, out=out)
    if bias is not None:
        y += bias
    return y


def fast_gemv(X, W, quant_state, out=None):
    """utils.py:872-977: out[1, 1, N] = X[1, 1, K] @ W^T for an NF4 weight (or a plain matmul when quant_state is None,
    :880-881). One uamd_gemv launch; the nested absmax is decoded inside it."""
    from . import decode as _dk
    _refuse_fp8([(W,)])
    if quant_state is None:
        return torch.matmul(X, W, out=out)
    N = int(quant_state.shape[0])
    flat = out.view(-1) if out is not None else None
    (y,) = _dk.linear_group(X.reshape(-1), [(W, quant_state, None, None, None)], out=flat)
    return y.view(1, 1, N)
