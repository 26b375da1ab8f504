"""Which kernels does hipBLASLt (torch.matmul) launch on the step's four GEMM shapes? Run under
`rocprofv3 --kernel-trace` and list the full Tensile kernel names (macro tile MT, MFMA instruction MI, wave
grid WG, ...) with tools/rocpd_names.py: the A/B reference for csrc/gemm256.hip."""
import torch

bf = torch.bfloat16
shapes = [(8192, 4096, 4096, "o"), (8192, 14336, 4096, "gate"), (8192, 4096, 14336, "down"), (8192, 4096, 6144, "qkv_dx"),
          (2048, 4096, 14336, "down@2k"), (2048, 14336, 4096, "gate@2k")]
for M, N, K, tag in shapes:
    X = torch.randn(M, K, device="cuda", dtype=bf)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    for _ in range(3):
        Y = X @ W.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        Y = X @ W.t()
    e.record()
    torch.cuda.synchronize()
    print(tag, M, N, K, round(2.0 * M * N * K * 10 / (s.elapsed_time(e) * 1e-3) / 1e12, 1), "TFLOP/s", flush=True)
