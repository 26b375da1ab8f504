"""uamd_lora_prepare: one launch = dtype copies (row-major + transposed) of every LoRA factor.
Oracle: torch's own `.to(dtype)` (round-to-nearest-even), which is what the reference does per use
(unsloth/kernels/utils.py:1166-1167). Bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_prepare_matches_torch_cast(dtype):
    from unsloth_amd.kernels import utils as U
    torch.manual_seed(0)
    shapes = [(16, 4096), (4096, 16), (1024, 16), (16, 14336), (14336, 16), (7, 33), (1, 8), (64, 64), (33, 1)]
    params = [torch.nn.Parameter(torch.randn(s, device="cuda") * 3) for s in shapes]
    U.invalidate_cast_cache()
    for P in params:                                   # first epoch: registered one by one
        assert torch.equal(U._cached_cast(P, "rowmajor", dtype, None), P.detach().to(dtype))
        assert torch.equal(U._cached_cast(P, "T", dtype, None), P.detach().to(dtype).t())
    with torch.no_grad():
        for P in params:
            P.mul_(1.7)
    U.invalidate_cast_cache()                          # next epoch: ONE launch refreshes all of them
    first = U._cached_cast(params[0], "rowmajor", dtype, None)
    torch.cuda.synchronize()
    for P in params:
        g = U._PREPARED[(P.device, dtype)].params[id(P)]
        assert torch.equal(g[1], P.detach().to(dtype)) and torch.equal(g[2], P.detach().to(dtype).t())
    assert first.data_ptr() == U._cached_cast(params[0], "rowmajor", dtype, None).data_ptr()
    with torch.no_grad():                              # in-place edit inside an epoch is picked up via _version
        params[3].add_(1.0)
    assert torch.equal(U._cached_cast(params[3], "T", dtype, None), params[3].detach().to(dtype).t())
