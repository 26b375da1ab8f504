"""One-shot environment probe on the GPU box: device, runtime library actually mapped, SDPA backends."""
import os
import time

import torch

print("torch", torch.__version__, "hip", torch.version.hip)
print("device", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
print("mem free/total GB", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info()])
from unsloth_amd import _lib
L = _lib.lib()
print("abi version", L.uamd_version())
print("hip libs mapped:", sorted({ln.split()[-1] for ln in open(f"/proc/{os.getpid()}/maps") if "amdhip" in ln}))
import torch.nn.functional as F
q = torch.randn(1, 32, 2048, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(1, 8, 2048, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(1, 8, 2048, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    try:
        with sdpa_kernel(be):
            o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
            o.sum().backward()
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(10):
                o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
                o.sum().backward()
            torch.cuda.synchronize()
            print(be, "ok fwd+bwd ms", (time.time() - t) / 10 * 1e3)
    except Exception as ex:
        print(be, "FAILED", type(ex).__name__, str(ex)[:200])
