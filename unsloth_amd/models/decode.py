"""KV-cache decode path; counterpart of the inference half of unsloth/models/llama.py.

Reference: LlamaAttention_fast_forward_inference (:352-569), fast_swiglu_inference (:572-606),
fast_rms_layernorm_inference (:609-632), LlamaModel_fast_forward_inference (:1249-1364) and unsloth_fast_generate
(:2167-2259): a Python loop of ~25 small torch / bitsandbytes calls per layer per token, with a KV cache that grows by
KV_CACHE_INCREMENT re-allocations and host syncs for the rotary length.

MI355X design: one `DecodeEngine` per (model, max_seq_len, batch):
  * the KV cache is allocated ONCE at [layers][B, Hk, S_max, D] (288 GB of HBM: no growth, no copies);
  * a token step of one sequence is FIVE launches per layer -- q|k|v GEMV (residual add + RMSNorm inside), attention (RoPE,
    cache append, split-KV attention and its combine inside), o GEMV, gate|up GEMV (residual add + RMSNorm in, SwiGLU out),
    down GEMV; every LoRA `A x` is computed once inside the GEMV launch that needs it (uamd_gemv_fused) -- that read the
    position from DEVICE memory, so the whole step -- all layers, final norm, lm_head GEMV, greedy argmax, position
    increment -- is captured once as a hipGraph and replayed per token. (Rounds 2-3: 14 launches per layer, ~450 kernels
    of ~9 us for ~0.6 ms of HBM traffic; still the batch > 1 path and `UNSLOTH_AMD_DECODE_FUSED=0`.);
  * prefill runs the training-path kernels (flash attention over the prompt) and writes K (post-RoPE) / V into the cache.
Batch > 1 decodes through the fused NF4 GEMM instead of the GEMV (as the reference does, utils.py:1095-1097).
"""
import math
import os

import torch

from ..kernels import decode as _dk
from ..kernels.rms_layernorm import add_rms_fwd, rms_fwd
from ..kernels.rope_embedding import fast_rope_embedding
from ..kernels.swiglu import swiglu_fg_kernel
from ..kernels.utils import get_lora_parameters_bias, matmul_lora

SPLIT_KEYS = 128


def _base(model):
    m = model.get_base_model() if hasattr(model, "get_base_model") else model
    return m


# UNSLOTH_AMD_DECODE_FUSED=0: one token of one sequence as the 14 separate launches per layer of rounds 2-3 instead of 5.
# (Round 3's first fused step -- 7 launches, every workgroup recomputing the norm AND t = A x, up to 48 rows x K from L2 --
# measured slower than the 14, 4.97 vs 4.13 ms per token, profiles/r03r_decode_fused_ab.jsonl; t is now computed once per
# launch and handed over through a device workspace, RoPE / append / combine ride in the attention launch and SwiGLU in the
# gate|up epilogue.)
FUSED_STEP = True


class DecodeEngine:
    """Greedy / sampled generation for a (PEFT-wrapped) Llama-family causal LM loaded by unsloth_amd."""

    def __init__(self, model, max_seq_len=2048, batch=1, use_graph=True):
        from . import llama as _ll
        self.model = model
        self.lm = _base(model)                    # LlamaForCausalLM
        self.core = self.lm.model                 # LlamaModel
        cfg = self.lm.config
        self.cfg = cfg
        self.dtype = _ll._model_dtype(self.core)
        self.dev = self.core.embed_tokens.weight.device
        self.B = batch
        self.S = int(math.ceil(max_seq_len / SPLIT_KEYS) * SPLIT_KEYS)
        self.Hq, self.Hk = cfg.num_attention_heads, cfg.num_key_value_heads
        self.D = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        if self.D != 128:
            raise NotImplementedError("decode attention kernel: head_dim 128 only")
        if getattr(cfg, "hidden_act", "silu") != "silu":
            raise NotImplementedError("decode path: SwiGLU MLP only")
        self.window = int(getattr(cfg, "sliding_window", None) or 0)
        self.scale = 1.0 / math.sqrt(self.D)
        self.use_graph = bool(use_graph) and batch == 1
        L = len(self.core.layers)
        kw = dict(dtype=self.dtype, device=self.dev)
        self.k_cache = [torch.zeros(batch, self.Hk, self.S, self.D, **kw) for _ in range(L)]
        self.v_cache = [torch.zeros(batch, self.Hk, self.S, self.D, **kw) for _ in range(L)]
        # [kv_len per sequence ..., step counter]: ONE add_(1) per token advances both. The step counter (never reset) is the
        # device half of the hand-off tags of the fused kernels (kernels/decode.HandOff): unlike the position it cannot repeat
        # when a new prompt is prefilled or a caller rewinds kv_len
        self._pos = torch.zeros(batch + 1, dtype=torch.int32, device=self.dev)
        self._pos[batch] = 1
        self.kv_len = self._pos[:batch]
        self.step_ctr = self._pos[batch:]
        self._steps = 1                               # host mirror of step_ctr (no sync): wrap-around guard
        self.partials = torch.empty(batch, self.Hq, self.S // SPLIT_KEYS, self.D + 2, dtype=torch.float32, device=self.dev)
        # the fused attention's splits: 128 keys each (64 keys = 256 workgroups at context 2048 measured 3.5 % slower per token,
        # profiles/r04_decode_split_keys_ab.txt), longer ones where 128 would mean more than 256 workgroups -- up to 256 of them
        # combine through granules, a larger launch pays the 21 us fence + arrival-counter tail: 256 keys at context 8192 with 8 KV
        # heads, 1024 at 32768
        self.fsplit = max(SPLIT_KEYS, int(math.ceil(self.S * self.Hk * batch / 256 / 16) * 16))
        self.fpartials, self.counters = _dk.fused_attn_workspace(batch, self.Hq, self.Hk, self.S, self.D, self.fsplit, self.dev,
                                                                  step_dev=self.step_ctr)
        self.sync = _dk.HandOff(self.dev, step_dev=self.step_ctr)                             # uamd_gemv_fused hand-off
        self.tok = torch.zeros(batch, 1, dtype=torch.long, device=self.dev)
        self.next_tok = torch.zeros(batch, dtype=torch.long, device=self.dev)
        self.logits = None
        tables = _ll._rope_tables(self.core)
        self.cos, self.sin = tables.get(self.S, self.dev, self.dtype)
        self._graph = None
        assert 5 * L + 1 < 1024, "hand-off tag sites: 5 per layer below UAMD_TAG_STRIDE"
        self._params = [self._layer_params(l) for l in self.core.layers]
        self._head = self.lm.lm_head.weight
        self._sample = None

    # ---- per-layer parameter tuples, read once (adapters must not be switched under a live engine) -------------------
    @staticmethod
    def _layer_params(layer):
        a, m = layer.self_attn, layer.mlp
        g = get_lora_parameters_bias
        return dict(qkv=[g(a.q_proj), g(a.k_proj), g(a.v_proj)], o=[g(a.o_proj)],
                    gu=[g(m.gate_proj), g(m.up_proj)], down=[g(m.down_proj)],
                    ln1=(layer.input_layernorm.weight, _eps(layer.input_layernorm)),
                    ln2=(layer.post_attention_layernorm.weight, _eps(layer.post_attention_layernorm)))

    # ---- prefill: prompt through the training-path kernels, K / V into the cache -----------------------------------------
    @torch.no_grad()
    def prefill(self, input_ids):
        from . import llama as _ll
        B, T = input_ids.shape
        assert B == self.B and T <= self.S
        # parameters were updated since the token step was captured (an optimizer stepped, a training forward ran): the
        # graph holds pointers to activation-dtype copies of the LoRA factors that are stale now -- capture again
        from ..kernels.utils import _CAST_EPOCH
        if getattr(self, "_epoch", None) != _CAST_EPOCH[0]:
            self._epoch = _CAST_EPOCH[0]
            # (a) the activation-dtype copies of the LoRA A factors are refreshed IN PLACE (the captured graph keeps
            # their addresses); (b) if a parameter's storage moved (optim.FlatAdamW adopts the parameters into its flat
            # arena, module.to(...)) the graph's pointers to the fp32 B factors dangle: capture again
            sig = []
            for P in self._params:
                for grp in ("qkv", "o", "gu", "down"):
                    As = [pr[2] for pr in P[grp] if pr[2] is not None]
                    if As:
                        _dk.lora_a_rows(As, self.dtype)
                    sig += [t.data_ptr() for pr in P[grp] for t in pr[:4] if torch.is_tensor(t)]
            sig = tuple(sig)
            if getattr(self, "_ptr_sig", None) != sig:
                self._graph = None
                self._ptr_sig = sig
        core = self.core
        h = core.embed_tokens(input_ids.to(self.dev)).to(self.dtype)
        cos, sin = self.cos, self.sin
        resid, delta = h, None
        for li, layer in enumerate(core.layers):
            w1, e1 = self._params[li]["ln1"]
            if delta is None:
                x = rms_fwd(resid, w1, e1)[0].view(B, T, -1)
            else:
                resid, x, _ = add_rms_fwd(delta, resid, w1, e1)
                resid, x = resid.view(B, T, -1), x.view(B, T, -1)
            attn = layer.self_attn
            Q, K, V = attn.apply_qkv(attn, x)
            Q = Q.view(B, T, self.Hq, self.D).transpose(1, 2)
            K = K.view(B, T, self.Hk, self.D).transpose(1, 2)
            V = V.view(B, T, self.Hk, self.D).transpose(1, 2)
            Q, K = fast_rope_embedding(Q, K, cos, sin, None)
            self.k_cache[li][:, :, :T].copy_(K)
            self.v_cache[li][:, :, :T].copy_(V)
            A = _ll._attention(Q, K, V, None, None, self.window or None)
            o = attn.apply_o(attn, A)
            w2, e2 = self._params[li]["ln2"]
            resid, x, _ = add_rms_fwd(o, resid, w2, e2)
            resid, x = resid.view(B, T, -1), x.view(B, T, -1)
            delta = layer.mlp(x)
        wn, en = core.norm.weight, _eps(core.norm)
        _, xn, _ = add_rms_fwd(delta[:, -1:], resid[:, -1:], wn, en)
        self.kv_len.fill_(T)
        self.logits = self._lm_head(xn.view(B, -1))
        return self.logits

    def _greedy(self):
        """next_tok = argmax of the step's logits, inside the captured step."""
        lg = self.logits
        if lg.dtype == torch.float32 and lg.is_contiguous():
            if getattr(self, "_amax_ws", None) is None:
                self._amax_ws = (torch.empty(self.B * 64, dtype=torch.float32, device=self.dev),
                                 torch.empty(self.B * 64, dtype=torch.long, device=self.dev))
            _dk.argmax_f32(lg, out=self.next_tok, ws=self._amax_ws)
        else:
            self.next_tok.copy_(torch.argmax(lg, dim=-1))

    def _lm_head(self, x):                       # x [B, H] -> fp32 logits [B, V]
        W = self._head
        if self.B == 1 and W.dtype == x.dtype and W.shape[1] <= 16384:
            (y,) = _dk.gemv(x.reshape(-1), [dict(W=W, N=W.shape[0], y_f32=True)], nf4=False)
            return y.view(1, -1)
        return (x @ W.t().to(x.dtype)).float()

    # ---- one token ----------------------------------------------------------------------------------------------------------
    def _linear(self, x, projs):
        """x [B, K] -> list of [B, N_i]."""
        if self.B == 1:
            return [y.view(1, -1) for y in _dk.linear_group(x, projs)]
        outs = []
        for p in projs:                           # batch > 1: the fused small-M NF4 GEMM (utils.py:1095-1097 does dequant + matmul)
            y = matmul_lora(x, p[0], p[1], p[2], p[3], p[4])
            if len(p) > 5 and p[5] is not None:
                y = y + p[5]
            outs.append(y)
        return outs

    @torch.no_grad()
    def _step_body(self):
        if self.B == 1 and FUSED_STEP:
            return self._step_body_fused()
        return self._step_body_plain()

    @torch.no_grad()
    def _step_body_fused(self):
        """One token of one sequence, 5 launches per decoder layer: q|k|v (residual add + RMSNorm inside), attention (RoPE,
        cache append and the split combine inside), o, gate|up (add + RMSNorm in, SwiGLU out), down; every LoRA `A x` inside
        the GEMV launch that consumes it."""
        H = self.cfg.hidden_size
        core = self.core
        resid = core.embed_tokens(self.tok).to(self.dtype).view(H)
        delta = None
        I = self.cfg.intermediate_size
        kw = dict(dtype=self.dtype, device=self.dev)
        nf4 = self._params[0]["gu"][0][1] is not None
        glu_ok = H <= (8192 if nf4 else 4096)            # a wave needs the gate row and the up row of an n in one trip
        for li in range(len(core.layers)):
            P = self._params[li]
            qkv = torch.empty(1, (self.Hq + 2 * self.Hk) * self.D, **kw)
            h = torch.empty_like(resid) if delta is not None else None
            site = 5 * li + 1                       # launch-site constants of the hand-off tags: 5 per layer
            _dk.linear_group(delta, P["qkv"], out=qkv.view(-1), fused=dict(mode=2, res=resid, norm_w=P["ln1"][0], eps=P["ln1"][1],
                                                                           h_out=h, sync=self.sync, site=site))
            if h is not None:
                resid = h
            a_out = torch.empty(1, self.Hq * self.D, **kw)
            _dk.attn_decode_fused(qkv, self.cos, self.sin, self.kv_len, self.k_cache[li], self.v_cache[li], a_out,
                                  self.fpartials, self.counters, self.fsplit, self.scale, self.Hq, window=self.window,
                                  site=site + 1)
            (o,) = _dk.linear_group(a_out.view(-1), P["o"], fused=dict(mode=0, sync=self.sync, site=site + 2))
            h = torch.empty_like(resid)
            pre = dict(mode=2, res=resid, norm_w=P["ln2"][0], eps=P["ln2"][1], h_out=h, sync=self.sync, site=site + 3)
            if glu_ok:
                (hmid,) = _dk.linear_group(o, P["gu"], fused=dict(pre, glu=True))
                (delta,) = _dk.linear_group(hmid, P["down"], fused=dict(mode=0, sync=self.sync, site=site + 4))
            else:
                gu = torch.empty(2 * I, **kw)
                _dk.linear_group(o, P["gu"], out=gu, fused=pre)
                (delta,) = _dk.linear_group(gu[:I], P["down"], fused=dict(mode=1, x2=gu[I:], sync=self.sync, site=site + 4))
            resid = h
        W = self._head
        if W.dtype == self.dtype and W.shape[1] <= 16384:
            (y,) = _dk.gemv(delta, [dict(W=W, N=W.shape[0], y_f32=True)], nf4=False,
                            pro=dict(mode=2, res=resid, norm_w=core.norm.weight, eps=_eps(core.norm), h_out=None))
            self.logits = y.view(1, -1)
        else:
            _, xn, _ = add_rms_fwd(delta.view(1, H), resid.view(1, H), core.norm.weight, _eps(core.norm))
            self.logits = self._lm_head(xn)
        self._pos.add_(1)                                   # positions and the step counter
        if self._sample is None:
            self._greedy()
        return self.logits

    @torch.no_grad()
    def _step_body_plain(self):
        B, H = self.B, self.cfg.hidden_size
        core = self.core
        h = core.embed_tokens(self.tok).to(self.dtype).view(B, H)
        resid, delta = h, None
        for li in range(len(core.layers)):
            P = self._params[li]
            if delta is None:
                x = rms_fwd(resid, *P["ln1"])[0]
            else:
                resid, x, _ = add_rms_fwd(delta, resid, *P["ln1"])
            if B == 1:
                qkv = torch.empty(1, (self.Hq + 2 * self.Hk) * self.D, dtype=self.dtype, device=self.dev)
                _dk.linear_group(x, P["qkv"], out=qkv.view(-1))
            else:
                qkv = torch.cat(self._linear(x, P["qkv"]), dim=1)
            _dk.rope_kv_append(qkv, self.cos, self.sin, self.kv_len, self.k_cache[li], self.v_cache[li],
                               self.Hq, self.Hk, self.D)
            a_out = torch.empty(B, self.Hq * self.D, dtype=self.dtype, device=self.dev)
            _dk.attn_decode(qkv[:, :self.Hq * self.D], self.k_cache[li], self.v_cache[li], self.kv_len, a_out,
                            self.partials, SPLIT_KEYS, self.scale, len_add=1, window=self.window)
            (o,) = self._linear(a_out, P["o"])
            resid, x, _ = add_rms_fwd(o, resid, *P["ln2"])
            gate, up = self._linear(x, P["gu"])
            hmid = swiglu_fg_kernel(gate.view(B, 1, -1), up.view(B, 1, -1)).view(B, -1)
            (delta,) = self._linear(hmid, P["down"])
        _, xn, _ = add_rms_fwd(delta, resid, core.norm.weight, _eps(core.norm))
        self.logits = self._lm_head(xn)
        self._pos.add_(1)                                   # positions and the step counter
        if self._sample is None:
            self._greedy()
        return self.logits

    def step(self, token_ids):
        """Feed one token per sequence ([B] or [B, 1]); returns fp32 logits [B, V] for the next position."""
        self.tok.copy_(token_ids.view(self.B, 1))
        self._steps += 1
        if self._steps >= (1 << 22) - 8:                     # step * TAG_STRIDE would wrap: start the tags over on clean workspaces
            self.sync.ws.zero_()
            self.fpartials.ws.zero_()
            self.step_ctr.fill_(1)
            self._steps = 1
        if not self.use_graph:
            return self._step_body()
        if self._graph is None:
            # eager warm-up (first-use attribute calls, allocator growth), with the position restored afterwards
            kv0 = self.kv_len.clone()
            self._step_body()
            self.kv_len.copy_(kv0)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._step_body()
            self._graph_logits = self.logits     # the graph's static output buffer (prefill() rebinds self.logits)
            self.kv_len.copy_(kv0)               # the capture itself does not execute
        self._graph.replay()
        self.logits = self._graph_logits
        return self.logits

    @torch.no_grad()
    def generate(self, input_ids, max_new_tokens=32, eos_token_id=None, do_sample=False, temperature=1.0, top_k=0,
                 generator=None, pad_token_id=None, top_p=1.0):
        """Greedy (default) or temperature / top-k / nucleus sampling. Returns [B, T + new] token ids. Rows that have produced
        an end-of-sequence token keep emitting `pad_token_id` (default: the first eos id), like HF's generate."""
        input_ids = input_ids.to(self.dev)
        logits = self.prefill(input_ids)
        out = [input_ids]
        eos = None if eos_token_id is None else sorted(set(eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id]))
        eos_t = None if eos is None else torch.tensor(eos, device=self.dev)
        pad = pad_token_id if pad_token_id is not None else (eos[0] if eos else 0)
        done = torch.zeros(self.B, dtype=torch.bool, device=self.dev)
        room = self.S - input_ids.shape[1]
        # a hand-off inside the fused decode launches that times out (csrc/decode.hip: a workgroup polled 2^18 times for a granule
        # that never came) poisons its output with NaN on purpose; NaN logits would otherwise turn into plausible-looking token ids
        # through argmax / multinomial. One element per row and token is accumulated (NaN * 0 = NaN) and looked at ONCE, after
        # the loop: no sync per token, and no silently wrong text.
        poison = torch.zeros(self.B, dtype=torch.float32, device=self.dev)
        for i in range(min(max_new_tokens, room)):
            poison += logits[:, 0].float() * 0.0
            if do_sample:
                lg = filter_logits(logits / max(temperature, 1e-6), top_k, top_p)
                nxt = torch.multinomial(torch.softmax(lg, dim=-1), 1, generator=generator).view(-1)
            else:
                nxt = torch.argmax(logits, dim=-1)
            if eos_t is not None:
                nxt = torch.where(done, torch.full_like(nxt, pad), nxt)      # finished rows: padding from here on
            out.append(nxt.view(self.B, 1))
            if eos_t is not None:
                done |= torch.isin(nxt, eos_t)
                if bool(done.all()):
                    break
            if i + 1 < min(max_new_tokens, room):
                logits = self.step(nxt)
        if bool(torch.isnan(poison).any()):
            raise RuntimeError("unsloth_amd decode: non-finite logits -- an in-launch hand-off of the fused decode step timed out "
                               "(not every workgroup of a launch was resident: a CU-masked / shared GPU?) or the model produced "
                               "NaN / inf. UNSLOTH_AMD_DECODE_FUSED=0 runs the 14-launch step without hand-offs.")
        return torch.cat(out, dim=1)


def filter_logits(logits, top_k=0, top_p=1.0):
    """HF's warper order and rules (generation/logits_process.py TopKLogitsWarper, TopPLogitsWarper) on [B, V] logits: keep the
    top_k largest (ties with the k-th stay), then the smallest set of most probable tokens whose mass reaches top_p -- a
    token goes when the cumulative probability of it and everything LESS likely is <= 1 - top_p; the most likely token
    always stays. Everything else becomes -inf."""
    if top_k and top_k < logits.shape[-1]:
        kth = torch.topk(logits, int(top_k), dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        ascending, order = torch.sort(logits, descending=False, dim=-1)
        drop = ascending.softmax(dim=-1).cumsum(dim=-1) <= (1.0 - float(top_p))
        drop[..., -1:] = False
        logits = logits.masked_fill(drop.scatter(-1, order, drop), float("-inf"))
    return logits


def _eps(norm):
    return float(getattr(norm, "variance_epsilon", getattr(norm, "eps", 1e-6)))


# generate() arguments the decode engine implements itself; anything else a caller passes changes what HF's generate would
# return, so such calls go to the original `generate` (kept as `model._old_generate` by for_inference) instead of being
# silently ignored
_ENGINE_KWARGS = {"max_new_tokens", "max_length", "eos_token_id", "pad_token_id", "do_sample", "temperature", "top_k", "top_p",
                  "attention_mask", "use_cache", "generator", "max_seq_len", "return_dict_in_generate", "output_scores",
                  "generation_config", "tokenizer"}


def unsloth_fast_generate(model, input_ids=None, max_new_tokens=None, max_seq_len=None, **kwargs):
    """llama.py:2167-2259 counterpart: `model.generate(...)` after `FastLanguageModel.for_inference(model)`.
    The reference forwards every keyword to HF's generate; the engine covers greedy / temperature / top-k / top-p decoding of
    unpadded batches. Calls outside that (repetition_penalty, beams, streamers, logits processors, a padded
    attention_mask, ...) are handed to HF's own generate (`model._old_generate`) -- or raise when it is unavailable."""
    input_ids = kwargs.pop("inputs", input_ids)
    attention_mask = kwargs.get("attention_mask", None)
    padded = attention_mask is not None and not bool(torch.all(attention_mask != 0))
    neutral = {"repetition_penalty": (None, 1.0), "num_beams": (None, 1), "streamer": (None,),
               "num_return_sequences": (None, 1), "min_new_tokens": (None, 0), "return_dict_in_generate": (None, False),
               "output_scores": (None, False), "generation_config": (None,)}
    unsupported = [k for k, v in kwargs.items()
                   if (k not in _ENGINE_KWARGS and not (k in neutral and v in neutral[k]))
                   or (k in ("return_dict_in_generate", "output_scores", "generation_config") and v)]
    if padded or unsupported:
        old = getattr(model, "_old_generate", None)
        if old is None:
            raise NotImplementedError(f"unsloth_fast_generate: unsupported arguments {unsupported or ['padded attention_mask']}")
        return old(input_ids, max_new_tokens=max_new_tokens, **kwargs) if max_new_tokens is not None else old(input_ids, **kwargs)
    gen_cfg = getattr(model, "generation_config", None)
    if gen_cfg is None:
        gen_cfg = getattr(_base(model), "generation_config", None)
    if max_new_tokens is None:
        max_length = kwargs.get("max_length", None)
        if max_length is not None:
            max_new_tokens = max(int(max_length) - input_ids.shape[1], 0)
        else:
            max_new_tokens = getattr(gen_cfg, "max_new_tokens", None) or 32
    eos = kwargs.get("eos_token_id", None)
    if eos is None and gen_cfg is not None:
        eos = getattr(gen_cfg, "eos_token_id", None)
    pad = kwargs.get("pad_token_id", None)
    if pad is None and gen_cfg is not None:
        pad = getattr(gen_cfg, "pad_token_id", None)
    # Sampling defaults come from the model's generation_config like in HF's generate (a checkpoint that ships
    # do_sample=True / temperature / top_k / top_p -- Llama-3-Instruct's does -- samples without the caller saying so);
    # explicit keywords win.
    def gen_default(name, fallback):
        if name in kwargs and kwargs[name] is not None:
            return kwargs[name]
        v = getattr(gen_cfg, name, None) if gen_cfg is not None else None
        return fallback if v is None else v
    do_sample = bool(gen_default("do_sample", False))
    temperature = float(gen_default("temperature", 1.0))
    top_k = int(gen_default("top_k", 0) or 0) if do_sample else 0
    top_p = float(gen_default("top_p", 1.0)) if do_sample else 1.0
    need = input_ids.shape[1] + max_new_tokens
    eng = getattr(model, "_uamd_decode_engine", None)
    if eng is None or eng.B != input_ids.shape[0] or eng.S < need:
        eng = DecodeEngine(model, max_seq_len=max(need, max_seq_len or 0), batch=input_ids.shape[0])
        model._uamd_decode_engine = eng
    return eng.generate(input_ids, max_new_tokens=max_new_tokens, eos_token_id=eos,
                        do_sample=do_sample, temperature=temperature, top_k=top_k, top_p=top_p,
                        generator=kwargs.get("generator", None), pad_token_id=pad)


# ---- the reference's function names, for code written against unsloth/models/llama.py ----------------------------------------
def fast_rms_layernorm_inference(self, X, XX=None, XX2=None, variance=None):
    """llama.py:609-632 (the scratch arguments are accepted and unused: one kernel, statistics in registers)."""
    return rms_fwd(X, self.weight, _eps(self))[0].view(X.shape)


def fast_swiglu_inference(self, X, temp_gate=None, temp_up=None, gate_multiplier=None, down_multiplier=None):
    """llama.py:572-606: down(silu(gate(X)) * up(X)) for one token per sequence; gate | up are one grouped launch."""
    from ..kernels.utils import fast_linear_forward
    bsz, q_len, _ = X.shape
    if bsz == 1 and q_len == 1:
        gate, up = _dk.linear_group(X.reshape(-1), [get_lora_parameters_bias(self.gate_proj),
                                                    get_lora_parameters_bias(self.up_proj)])
        gate, up = gate.view(1, 1, -1), up.view(1, 1, -1)
    else:
        gate, up = fast_linear_forward(self.gate_proj, X), fast_linear_forward(self.up_proj, X)
    if gate_multiplier is not None:
        gate = gate * gate_multiplier
    h = swiglu_fg_kernel(gate, up)
    down = fast_linear_forward(self.down_proj, h)
    return down * down_multiplier if down_multiplier is not None else down


def LlamaAttention_fast_forward_inference(self, hidden_states, past_key_value, position_ids, do_prefill=False,
                                          attention_mask=None, rotary_seq_len=None):
    """llama.py:352-569: one new token per sequence against the KV cache. `past_key_value` = (K, V) [B, Hk, S, D] seeds the
    module's cache when `do_prefill` (allocated once at the model's max_seq_length instead of growing by
    KV_CACHE_INCREMENT); returns (attention output [B, 1, hidden], (K, V) views of the cache incl. the new token).
    Key-padding masks are not supported here (the training path handles them through SDPA)."""
    from . import llama as _ll
    from ..kernels.utils import fast_linear_forward
    if attention_mask is not None:
        raise NotImplementedError("decode attention takes no padding mask; left-pad-free batches only")
    cfg = self.config
    bsz = hidden_states.shape[0]
    Hq, Hk, D = cfg.num_attention_heads, cfg.num_key_value_heads, self.head_dim
    dev, dtype = hidden_states.device, hidden_states.dtype
    K1, V1 = past_key_value
    seq_len = K1.shape[-2]
    st = getattr(self, "_uamd_kv", None)
    if do_prefill or st is None or st["k"].shape[0] != bsz:
        S = int(math.ceil(max(getattr(self, "max_seq_length", 0) or 0, seq_len + 1, 2048) / SPLIT_KEYS) * SPLIT_KEYS)
        st = dict(k=torch.zeros(bsz, Hk, S, D, dtype=dtype, device=dev), v=torch.zeros(bsz, Hk, S, D, dtype=dtype, device=dev),
                  len=torch.zeros(bsz, dtype=torch.int32, device=dev),
                  part=torch.empty(bsz, Hq, S // SPLIT_KEYS, D + 2, dtype=torch.float32, device=dev))
        st["k"][:, :, :seq_len].copy_(K1)
        st["v"][:, :, :seq_len].copy_(V1)
        self._uamd_kv = st
    st["len"].fill_(seq_len)
    qkv = torch.cat([fast_linear_forward(p, hidden_states).view(bsz, -1) for p in (self.q_proj, self.k_proj, self.v_proj)], dim=1) \
        if bsz != 1 else torch.empty(1, (Hq + 2 * Hk) * D, dtype=dtype, device=dev)
    if bsz == 1:
        _dk.linear_group(hidden_states.reshape(-1), [get_lora_parameters_bias(p) for p in (self.q_proj, self.k_proj, self.v_proj)],
                         out=qkv.view(-1))
    if position_ids.dim() == 1:
        position_ids = position_ids[:, None]
    pos = position_ids[:, -1].to(device=dev, dtype=torch.int32).contiguous()
    tables = getattr(self, "_uamd_rope", None)
    if tables is None:
        tables = self._uamd_rope = _ll.RopeTables(cfg)
    cos, sin = tables.get(max(seq_len + 1, st["k"].shape[2]), dev, dtype)
    _dk.rope_kv_append(qkv, cos, sin, st["len"], st["k"], st["v"], Hq, Hk, D, rope_pos=pos)
    out = torch.empty(bsz, Hq * D, dtype=dtype, device=dev)
    _dk.attn_decode(qkv[:, :Hq * D], st["k"], st["v"], st["len"], out, st["part"], SPLIT_KEYS, 1.0 / math.sqrt(D),
                    len_add=1, window=int(getattr(cfg, "sliding_window", None) or 0))
    A = fast_linear_forward(self.o_proj, out.view(bsz, 1, Hq * D))
    return A, (st["k"][:, :, :seq_len + 1], st["v"][:, :, :seq_len + 1])
