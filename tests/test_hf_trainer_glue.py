"""CPU: the glue between a patched model and HuggingFace's STOCK `transformers.Trainer` (SURVEY 8 a20 + 9.9; ref
models/_utils.py:187-249, 3142-3313, loader.py:1114, llama.py:2988-3000, 3228, 3575-3595).

The HIP kernels cannot run here, so the two COMPUTE entry points of the patched CausalLM forward (the decoder stack and the
fused linear cross-entropy) are replaced by small torch stand-ins; everything else is the product: the PEFT-style wrapper,
PeftModel_fast_forward, the CausalLM forward's argument plumbing, the Trainer patches. tests/test_gpu_hf_trainer.py runs
the same drive on the real kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._hf_trainer_util import Docs, PaddedCollator, PaddingFreeCollator, replay, run_stock_trainer, shifted_targets


@pytest.fixture(scope="module", autouse=True)
def _restore_class_patches():
    """from_pretrained() patches the HF classes for the process (llama.py:2288-2320); other test modules start from stock classes."""
    yield
    from unsloth_amd.kernels import unpatch_rms_layernorm
    from unsloth_amd.kernels.cross_entropy_loss import unpatch_loss_functions
    from unsloth_amd.models import llama as L
    L.unpatch_all()
    unpatch_rms_layernorm()
    unpatch_loss_functions()


def _tiny_cpu_model(seed=3407):
    from transformers import LlamaConfig
    from unsloth_amd import FastLanguageModel
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=16, vocab_size=100, max_position_embeddings=128,
                      tie_word_embeddings=False)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=64, dtype=torch.bfloat16, load_in_4bit=False,
                                                 device="cpu", random_state=seed, use_gradient_checkpointing=True)
    model = FastLanguageModel.get_peft_model(model, r=4, lora_alpha=8, use_gradient_checkpointing=True, random_state=seed)
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return model


@pytest.fixture()
def torch_compute(monkeypatch):
    """Stand-ins for the two compute entry points; records the keyword arguments the model forward received."""
    from unsloth_amd.models import llama as L
    calls = []

    def stack(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, **kwargs):
        calls.append(dict(kwargs, position_ids=position_ids))
        h = self.embed_tokens(input_ids).float()
        for layer in self.layers:
            q = layer.self_attn.q_proj
            ad = q.active_adapters[0]
            h = h + (h @ q.lora_A[ad].weight.t() @ q.lora_B[ad].weight.t()) * q.scaling[ad]
        return h

    def fused_ce(trainer, hidden_states, lm_head_weight, lm_head_bias, labels, mask=None, n_items=None, **kw):
        logits = hidden_states.float() @ lm_head_weight.float().t()
        total = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1),
                                                  ignore_index=-100, reduction="sum")
        return total / n_items

    monkeypatch.setattr(L, "LlamaModel_fast_forward", stack)
    monkeypatch.setattr(L, "unsloth_fused_ce_loss", fused_ce)
    return calls


@pytest.mark.parametrize("collator_kind,bf16", [("padding_free", False), ("padded", False), ("padding_free", True)])
def test_stock_trainer_drives_the_patched_model_with_window_wide_shifted_num_items(tmp_path, torch_compute, collator_kind, bf16):
    """bf16=True: accelerate wraps the forward in autocast + an fp32 conversion that walks every field of the model output --
    EMPTY_LOGITS included (it must survive being LOOKED at)."""
    model = _tiny_cpu_model()
    data = Docs(40, 100, 5, 14)
    collator = PaddingFreeCollator() if collator_kind == "padding_free" else PaddedCollator(12)
    losses, trainer = run_stock_trainer(model, data, collator, tmp_path, use_cpu=True, bf16=bf16)
    assert len(losses) == 3 and trainer.state.global_step == 3
    assert trainer.model_accepts_loss_kwargs
    used = collator.seen[:6]
    assert len(torch_compute) == 6
    for step in range(3):
        want = sum(shifted_targets(b) for b in used[2 * step:2 * step + 2])       # the WHOLE window, shifted labels
        for call in torch_compute[2 * step:2 * step + 2]:
            assert int(call["num_items_in_batch"]) == want
    if collator_kind == "padding_free":
        assert all(c["packed_seq_lengths"] is not None and c["position_ids"] is not None for c in torch_compute)
    else:
        # labels == ids on [3, 12] rows: the unshifted count would be 3 per micro-batch too many
        assert int(torch_compute[0]["num_items_in_batch"]) == 2 * 3 * 11
    # the same six batches through a hand-written accumulation loop on a twin model: the same losses
    twin = _tiny_cpu_model()
    opt = torch.optim.AdamW([p for p in twin.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01)
    want = replay(twin, used, "cpu", optimizer=opt, autocast=bf16)
    assert losses == pytest.approx(want, abs=2e-4), (losses, want)          # Trainer rounds its log to 4 decimals
    assert losses[0] > losses[-1] - 0.5                                       # trains, does not diverge


def test_marker_keeps_trainer_from_wrapping_in_data_parallel():
    """models/_utils.py:187-241: `Trainer._wrap_model` with n_gpu == 2 -- nn.DataParallel around an unmarked model, the
    marked one comes back as it went in, and `args._n_gpu` is what it was."""
    from transformers.trainer import Trainer
    from unsloth_amd.models._utils import mark_disable_data_parallel, patch_trainer_data_parallel
    assert patch_trainer_data_parallel() and patch_trainer_data_parallel()            # idempotent
    assert Trainer._wrap_model._unsloth_data_parallel_patched

    class Args:
        _n_gpu = 2
        n_gpu = property(lambda self: self._n_gpu)

    class Accel:
        @staticmethod
        def unwrap_model(m, keep_torch_compile=False):
            return m

    class Self:
        args = Args()
        accelerator = Accel()
        is_fsdp_xla_enabled = False

    plain, marked = torch.nn.Linear(2, 2), mark_disable_data_parallel(torch.nn.Linear(2, 2))
    assert isinstance(Trainer._wrap_model(Self(), plain), torch.nn.DataParallel)
    assert Trainer._wrap_model(Self(), marked) is marked
    assert Self.args._n_gpu == 2
    model = _tiny_cpu_model()
    assert model._unsloth_disable_data_parallel and model.get_base_model()._unsloth_disable_data_parallel
    assert Trainer._wrap_model(Self(), model) is model


def test_load_path_lists_the_rotary_buffers_for_ddp():
    """loader.py:1114, llama.py:3228 / 3595: from_pretrained and get_peft_model (first call and idempotent re-call) leave the
    rotary buffers' CURRENT fully qualified names on the object DDP will be given."""
    from unsloth_amd import FastLanguageModel
    model = _tiny_cpu_model()
    names = {n for n, _ in model.named_buffers() if n.rsplit(".", 1)[-1] in ("inv_freq", "original_inv_freq")}
    assert names and names <= set(model._ddp_params_and_buffers_to_ignore)
    again = FastLanguageModel.get_peft_model(model, r=4, lora_alpha=8)
    assert again is model and names <= set(again._ddp_params_and_buffers_to_ignore)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _tiny_cpu_model()
        rot = model.get_base_model().model.rotary_emb
        rot.register_buffer("synced_probe", torch.zeros(3), persistent=False)         # NOT on the ignore list
        with torch.no_grad():
            rot.inv_freq.fill_(100.0 + rank)            # the per-replica buffer DDP must leave alone
            rot.synced_probe.fill_(7.0 + rank)
        ddp = torch.nn.parallel.DistributedDataParallel(model)      # construction broadcasts rank 0's state
        ignored = set(ddp.parameters_to_ignore)
        q.put((rank, float(rot.inv_freq[0]), float(rot.synced_probe[0]),
               any(n.endswith("rotary_emb.inv_freq") for n in ignored),
               sorted(n for n, _ in ddp.module.named_parameters() if _.requires_grad)[:1]))
    finally:
        dist.destroy_process_group()


def test_world_2_gloo_ddp_honours_the_ignore_list():
    """A user who wraps the patched model in torch DDP anyway: every buffer is broadcast from rank 0 at construction --
    except the rotary inv_freq buffers the load path put on `_ddp_params_and_buffers_to_ignore` (on NCCL a CPU-resident
    inv_freq would otherwise crash the broadcast, loader_utils.py:834-848)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, inv0, probe0, ign0, _), (r1, inv1, probe1, ign1, _) = got
    assert ign0 and ign1
    assert (inv0, inv1) == (100.0, 101.0)             # untouched on both ranks
    assert (probe0, probe1) == (7.0, 7.0)             # an ordinary buffer WAS synchronised from rank 0
