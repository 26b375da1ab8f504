// How densely can ONE wave per SIMD issue independent v_mfma_f32_16x16x32_bf16 back to back, against TWO waves per SIMD?
// (round 5: the 4-wave x 128x128 form of the GEMM -- hipBLASLt's decomposition -- measured 16 % slower than the shipped 8-wave
// kernel "with nothing else in the loop"; is that the hardware or the build?)
//   Each wave runs ITER trips of NM independent MFMAs (NM distinct accumulator tuples, A / B operands constant), clock64 around the
//   loop; reported: matrix-pipe cycles per MFMA per SIMD (16 = the pipe's own rate) and the chip's TFLOP/s at that density.
//   variants: waves per SIMD 1 / 2; accumulators in VGPRs ("+v") or AGPRs ("+a"); optionally FILL non-MFMA instructions (s_nop 0)
//   between consecutive MFMAs (what a real K loop threads through the MFMA stream).
//   hipcc --offload-arch=gfx950 -O3 -w tools/probes/mfma_issue_probe.hip -o tools/probes/build/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NM, bool AGPR, int FILL, int WPS>
__global__ void __launch_bounds__(256 * WPS) probe(float* out, long long* cyc, int iters) {
    f32x4 acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (FILL >= 1) asm volatile("s_nop 0");
            if (FILL >= 2) asm volatile("s_nop 0");
            if (FILL >= 3) asm volatile("s_nop 0");
            if (FILL >= 4) asm volatile("s_nop 0\n\ts_nop 0");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, bool AGPR, int FILL, int WPS>
void run(const char* name, float* out, long long* cyc) {
    const int waves_per_simd = WPS;
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<NM, AGPR, FILL, WPS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NM, AGPR, FILL, WPS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256.0;                                                   // clock64 ticks (100 MHz REFCLK on gfx9: report wall time too)
    const double mfma_per_simd = (double)iters * NM * waves_per_simd;
    const double flop = 256.0 * 4 * mfma_per_simd * 16 * 16 * 32 * 2;
    printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"mfma_per_trip\": %d, \"agpr\": %d, \"fill\": %d, \"ms\": %.3f, \"TFLOPs\": %.0f, "
           "\"ns_per_mfma_per_simd\": %.2f}\n", name, waves_per_simd, NM, (int)AGPR, FILL, ms, flop / (ms * 1e-3) / 1e12,
           ms * 1e6 / mfma_per_simd);
    fflush(stdout);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    run<32, false, 0, 1>("1w_vgpr32", out, cyc);
    run<32, false, 0, 2>("2w_vgpr32", out, cyc);
    run<64, true, 0, 1>("1w_agpr64", out, cyc);
    run<32, true, 0, 2>("2w_agpr32", out, cyc);
    run<64, true, 1, 1>("1w_agpr64_fill1", out, cyc);
    run<64, true, 2, 1>("1w_agpr64_fill2", out, cyc);
    run<64, true, 3, 1>("1w_agpr64_fill3", out, cyc);
    run<64, true, 4, 1>("1w_agpr64_fill5", out, cyc);
    run<32, true, 1, 2>("2w_agpr32_fill1", out, cyc);
    run<32, true, 2, 2>("2w_agpr32_fill2", out, cyc);
    run<32, true, 3, 2>("2w_agpr32_fill3", out, cyc);
    run<32, true, 4, 2>("2w_agpr32_fill5", out, cyc);
    run<32, false, 1, 1>("1w_vgpr32_fill1", out, cyc);
    run<32, false, 2, 1>("1w_vgpr32_fill2", out, cyc);
    run<32, false, 3, 1>("1w_vgpr32_fill3", out, cyc);
    run<32, false, 4, 1>("1w_vgpr32_fill5", out, cyc);
    run<16, false, 0, 1>("1w_vgpr16", out, cyc);
    run<8, false, 0, 1>("1w_vgpr8", out, cyc);
    run<4, false, 0, 1>("1w_vgpr4", out, cyc);
    return 0;
}
