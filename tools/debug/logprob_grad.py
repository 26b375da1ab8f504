"""Bisect of the config-5 log-prob gradient mismatch: kernel-level d(hidden) of chunked_hidden_states_selective_log_softmax
at the failing shape (T = 4095 rows in 4 chunks of 1024 / 1023, V = 32000, H = 4096) against torch autograd in fp32."""
import sys
import torch
sys.path.insert(0, ".")
from unsloth_amd.models.rl_replacements import chunked_hidden_states_selective_log_softmax as f

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (T, H, V, chunks) in [(4095, 4096, 32000, 4), (4096, 4096, 32000, 4), (4095, 4096, 32000, 1), (1023, 4096, 32000, 1),
                          (1024, 4096, 32000, 1), (1024, 1024, 32000, 1), (1024, 4096, 8192, 1)]:
    h = (torch.randn(1, T, H, generator=g) * 0.5).to(torch.bfloat16).to(dev).requires_grad_(True)
    W = (torch.randn(V, H, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    idx = torch.randint(0, V, (1, T), generator=g).to(dev)
    w = torch.randn(1, T, generator=g).to(dev)
    w[:, : T // 2] = 0
    lp = f(h, W, idx, chunks=chunks)
    (lp * w).sum().backward()
    got = h.grad.float()
    h32 = h.detach().float().requires_grad_(True)
    ref_lp = torch.log_softmax(h32 @ W.float().t(), dim=-1).gather(-1, idx.unsqueeze(-1)).squeeze(-1)
    (ref_lp * w).sum().backward()
    ref = h32.grad
    rel = ((got - ref).norm() / ref.norm()).item()
    per_chunk = []
    cr = min(4096, max(256, -(-T // chunks)))
    for r0 in range(0, T, cr):
        a, b = got[0, r0:r0 + cr], ref[0, r0:r0 + cr]
        per_chunk.append(round(((a - b).norm() / (b.norm() + 1e-30)).item(), 4))
    print(f"T={T} H={H} V={V} chunks={chunks}: lp err {(lp - ref_lp).abs().max().item():.4f}  dh rel {rel:.4f}  per chunk {per_chunk}",
          flush=True)
