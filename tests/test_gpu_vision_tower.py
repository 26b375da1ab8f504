"""-m gpu: Qwen2-VL's vision blocks on the hand kernels (models/vision_tower.py: 2-D RoPE through csrc/rope_embedding.hip,
non-causal attention inside the `cu_seqlens` windows through csrc/attention.hip, QuickGELU through csrc/glu.hip) against the SAME
transformers modules run in fp32 with their own forward (eager attention, torch QuickGELU) -- the implementation-independent
oracle of this path (the reference's VLM path is the unsloth_zoo compiler over these modules, unsloth/models/vision.py:881-1990).
Real widths of Qwen2-VL-7B's ViT: 1280 wide, 16 heads of 80, MLP 5120, patch 14, merge 2."""
import copy

import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tower(depth=2):
    from transformers import Qwen2VLConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
    cfg = Qwen2VLConfig(vision_config=dict(depth=depth, embed_dim=1280, hidden_size=3584, num_heads=16, mlp_ratio=4, patch_size=14,
                                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3)).vision_config
    cfg._attn_implementation = "eager"
    torch.manual_seed(7)
    vis = Qwen2VisionTransformerPretrainedModel._from_config(cfg) if hasattr(Qwen2VisionTransformerPretrainedModel, "_from_config") \
        else Qwen2VisionTransformerPretrainedModel(cfg)
    g = torch.Generator().manual_seed(8)
    for p in vis.parameters():                     # LayerNorm at (1, 0) and zero biases would hide mistakes
        if p.dim() == 1:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.1 + (1.0 if p.mean() > 0.5 else 0.0))
    return vis


def test_quick_gelu_kernel_matches_torch_fp32():
    from unsloth_amd.kernels.quick_gelu import fast_quick_gelu
    for dtype, n in ((torch.bfloat16, (4096, 5120)), (torch.float16, (37, 1001))):
        x = (torch.randn(*n, generator=torch.Generator().manual_seed(1)) * 2).to(dtype)
        dy = torch.randn(*n, generator=torch.Generator().manual_seed(2)).to(dtype)
        xr = x.float().requires_grad_(True)
        yr = xr * torch.sigmoid(1.702 * xr)
        yr.backward(dy.float())
        xd = x.to(DEV).requires_grad_(True)
        y = fast_quick_gelu(xd)
        y.backward(dy.to(DEV))
        ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
        assert (y.float().cpu() - yr.detach()).abs().max() <= ulp * max(1.0, yr.abs().max().item())
        assert (xd.grad.float().cpu() - xr.grad).abs().max() <= 2 * ulp * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("grids", [[(1, 32, 32)], [(1, 16, 24), (1, 32, 20), (2, 8, 8)]], ids=["one_image", "three_windows"])
def test_vision_tower_on_hand_kernels_matches_transformers_fp32(grids):
    """Output of the whole tower (patch-embed -> blocks -> merger) and the gradient reaching pixel_values, bf16 on the hand kernels
    against fp32 transformers; the patched forwards are really taken (attention launches counted, no SDPA / eager call)."""
    from unsloth_amd.kernels import attention as flash
    from unsloth_amd.kernels.layernorm import patch_layernorm
    from unsloth_amd.models.vision_tower import patch_vision_tower
    ref = _tower().to(DEV).float()
    fast = copy.deepcopy(ref).to(torch.bfloat16)
    for m_ in (ref, fast):                                           # the oracle sees the bf16-rounded weights
        pass
    ref.load_state_dict({k: v.float() for k, v in fast.state_dict().items()})
    patch_layernorm()
    assert patch_vision_tower(fast) == 2
    thw = torch.tensor(grids, device=DEV)
    n_patches = int(sum(t * h * w for t, h, w in grids))
    g = torch.Generator().manual_seed(3)
    pix = torch.randn(n_patches, 3 * 2 * 14 * 14, generator=g).to(DEV)
    calls = []
    real = flash.attn_forward
    flash.attn_forward = lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1]
    try:
        pf = pix.to(torch.bfloat16).requires_grad_(True)
        out_f = fast(pf, grid_thw=thw)
        out_f = out_f.pooler_output if hasattr(out_f, "pooler_output") else out_f
    finally:
        flash.attn_forward = real
    assert len(calls) == 2 and all(s == (1, n_patches, 16, 80) for s in calls), calls
    pr = pix.to(torch.bfloat16).float().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):          # the oracle's Conv3d without MIOpen's minutes of kernel search
        out_r = ref(pr, grid_thw=thw)
        out_r = out_r.pooler_output if hasattr(out_r, "pooler_output") else out_r
        w = torch.randn(out_r.shape, generator=g).to(DEV)
        (out_r * w).sum().backward()
    (out_f.float() * w).sum().backward()
    assert rel_fro(out_f.float().cpu(), out_r.detach().cpu()) < 1.5e-2
    assert rel_fro(pf.grad.float().cpu(), pr.grad.cpu()) < 3e-2
