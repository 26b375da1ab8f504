"""-m gpu: optim.FlatAdamW (csrc/adamw.hip: one launch over flat parameter / gradient / moment arenas) against
torch.optim.AdamW's single-tensor reference implementation."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


class _Bag(torch.nn.Module):
    """Parameters with LoRA-like names and shapes (every numel a multiple of 4, like r x in / out x r factors)."""

    def __init__(self, shapes, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.layers = torch.nn.ModuleList()
        for i, (a, b) in enumerate(shapes):
            m = torch.nn.Module()
            m.lora_A = torch.nn.Parameter(torch.randn(a, generator=g).to(DEV))
            m.lora_B = torch.nn.Parameter(torch.randn(b, generator=g).to(DEV))
            self.layers.append(m)


@pytest.mark.parametrize("wd,lr", [(0.01, 2e-4), (0.0, 1e-2), (0.1, 5e-3)])
def test_flat_adamw_matches_torch_adamw(wd, lr):
    from unsloth_amd.optim import FlatAdamW
    shapes = [((16, 64), (64, 16)), ((8, 1000), (36, 8)), ((4, 4), (1024, 16))]
    ref_model = _Bag(shapes)
    model = copy.deepcopy(ref_model)
    ref = torch.optim.AdamW(ref_model.parameters(), lr=lr, weight_decay=wd, betas=(0.9, 0.999), foreach=False, fused=False)
    opt = FlatAdamW(model, lr=lr, weight_decay=wd)
    rp = dict(ref_model.named_parameters())
    for step in range(6):
        gen = torch.Generator().manual_seed(50 + step)
        for n, p in model.named_parameters():
            gr = (torch.randn(p.shape, generator=gen) * (0.05 + 0.3 * step)).to(DEV)
            rp[n].grad = gr.clone()
            assert p.grad is not None and p.grad.data_ptr() == opt.arena._views[id(p)].data_ptr()
            p.grad.add_(gr)                       # gradients are ADDED into the (zeroed) arena, like uamd_lora_tn does
            opt.arena.ready(p)
        if step == 3:
            for grp in opt.param_groups:          # an LR scheduler writes param_groups
                grp["lr"] = lr * 0.5
            for grp in ref.param_groups:
                grp["lr"] = lr * 0.5
        ref.step()
        opt.step()
        opt.zero_grad()
        ref.zero_grad(set_to_none=True)
        assert float(opt.arena.arena.abs().max()) == 0.0          # zeroed by the step's own pass
    for n, p in model.named_parameters():
        # torch's single-tensor path updates exp_avg by lerp_ (m + (1 - b1)(g - m)), the kernel by b1 m + (1 - b1) g like
        # torch's fused path: the two differ by a few ulp of the LARGER term, hence absolute bounds at the operands' scale
        # (|g| up to ~6 here; measured worst differences 1.5e-8 on exp_avg, 2.4e-7 on the parameters)
        torch.testing.assert_close(p.data, rp[n].data, rtol=1e-5, atol=1e-6)
        st, rst = opt.state[p], ref.state[rp[n]]
        torch.testing.assert_close(st["exp_avg"], rst["exp_avg"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(st["exp_avg_sq"], rst["exp_avg_sq"], rtol=1e-5, atol=1e-10)
        assert int(st["step"]) == 6
        assert p.data_ptr() >= opt.flat_p.data_ptr() and p.data_ptr() < opt.flat_p.data_ptr() + opt.flat_p.numel() * 4


def test_flat_adamw_grad_scale_discarded_grads_and_state_dict():
    from unsloth_amd.optim import FlatAdamW
    shapes = [((16, 32), (32, 16))]
    a, b = _Bag(shapes, seed=3), _Bag(shapes, seed=3)
    oa, ob = FlatAdamW(a, lr=1e-2), FlatAdamW(b, lr=1e-2)
    gen = torch.Generator().manual_seed(9)
    grads = [(torch.randn(p.shape, generator=gen)).to(DEV) for p in a.parameters()]
    # (1) gradients thrown away without a step: zero_grad() must really clear the arena
    for p, gr in zip(a.parameters(), grads):
        p.grad.add_(gr * 7)
        oa.arena.ready(p)
    oa.zero_grad()
    assert float(oa.arena.arena.abs().max()) == 0.0
    # (2) step(grad_scale = c) on g  ==  step() on c * g
    for (p, q), gr in zip(zip(a.parameters(), b.parameters()), grads):
        p.grad.add_(gr)
        oa.arena.ready(p)
        q.grad.add_(gr * 0.25)
        ob.arena.ready(q)
    oa.step(grad_scale=0.25)
    ob.step()
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(p.data, q.data, rtol=1e-6, atol=1e-8)
    # (3) state_dict round trip into a fresh optimizer keeps stepping identically
    c = _Bag(shapes, seed=3)
    with torch.no_grad():
        for p, q in zip(c.parameters(), a.parameters()):
            p.copy_(q)
    oc = FlatAdamW(c, lr=1e-2)
    oc.load_state_dict(oa.state_dict())
    for (p, q), gr in zip(zip(a.parameters(), c.parameters()), grads):
        p.grad.add_(gr)
        oa.arena.ready(p)
        q.grad.add_(gr)
        oc.arena.ready(q)
    oa.step()
    oc.step()
    for p, q in zip(a.parameters(), c.parameters()):
        assert torch.equal(p.data, q.data)
        assert oc.state[q]["exp_avg"].data_ptr() >= oc.flat_m.data_ptr()


def test_training_with_flat_adamw_tracks_torch_adamw():
    """Whole path: the tiny QLoRA model trained for a few steps with make_optimizer's FlatAdamW and with torch's AdamW
    from the same start -- same loss curve (the optimizers differ by rounding only), parameters live in the flat arena,
    the cached bf16 copies of the factors follow them (kernels/utils._PreparedFactors reads the new storage)."""
    from tests.test_gpu_model import _batch, _tiny
    from unsloth_amd.optim import FlatAdamW
    from unsloth_amd.trainer import make_optimizer, training_step
    ids, labels, pos = _batch(B=2, T=64, seed=4)
    batch = dict(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    curves = []
    for flat in (True, False):
        model = _tiny(r=16, gc=False, head_dim=128)
        opt = make_optimizer(model, lr=2e-3, flat=flat)
        assert isinstance(opt, FlatAdamW) == flat
        curves.append([float(training_step(model, batch, opt)) for _ in range(8)])
        if flat:
            assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in model.parameters() if p.requires_grad)
            opt.close()
    assert curves[0][-1] < curves[0][0] - 0.5                       # it learns
    for x, y in zip(*curves):
        assert abs(x - y) <= 2e-3 * abs(y), curves
