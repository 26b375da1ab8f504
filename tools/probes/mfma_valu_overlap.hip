// Does ONE wave per SIMD overlap its own MFMAs with its own VALU work on gfx950? (attn_fwd64_kernel's premise.)
// 256 blocks x 256 threads (one wave per SIMD, launch_bounds + 128 KiB of LDS keep a second block off the CU). Per
// iteration: NM independent v_mfma_f32_32x32x16_bf16 (4 accumulators) and NV independent VALU ops (v_fma / v_exp).
//   mode 0: MFMA only   1: VALU only   2: both, one MFMA then NV/NM VALU, repeated   3: both, all MFMAs then all VALU
//   modes 4-7: the same with TWO waves per SIMD (512 threads): what the 8-wave kernels get from each other
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo     (result: profiles/r02_mfma_valu_overlap.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int MODE, int EXP>
__global__ void __launch_bounds__(512) probe(float* out, int iters) {
    extern __shared__ unsigned char smem[];
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i * 0.01f); }
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float c0 = 1.0001f, c1 = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MODE == 0 || MODE == 2) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {                    // 8 VALU ops per MFMA (32 cycles of plain VALU)
                    const int k = (m * 8 + j) & 15;
                    if (EXP && (j & 3) == 0) v[k] = __builtin_amdgcn_exp2f(v[k] * c1);
                    else v[k] = __builtin_fmaf(v[k], c0, c1);
                }
            }
            if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 128; ++m) {
                const int k = m & 15;
                if (EXP && (m & 3) == 0) v[k] = __builtin_amdgcn_exp2f(v[k] * c1);
                else v[k] = __builtin_fmaf(v[k], c0, c1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int EXP>
float run(int threads, float* out, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, EXP><<<256, threads, 128 * 1024>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, EXP><<<256, threads, 128 * 1024>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;                                  // 32,000 MFMAs and 256,000 VALU ops per wave
    for (int threads = 256; threads <= 512; threads *= 2) {
        printf("%d wave(s) per SIMD, %d iterations of {16 MFMA 32x32x16, 128 VALU}\n", threads / 256, iters);
        printf("  fma only : mfma %.0f us | valu %.0f us | interleaved 1:8 %.0f us | 16 then 128 %.0f us\n",
               run<0, 0>(threads, out, iters), run<1, 0>(threads, out, iters), run<2, 0>(threads, out, iters), run<3, 0>(threads, out, iters));
        printf("  1/4 v_exp: mfma %.0f us | valu %.0f us | interleaved 1:8 %.0f us | 16 then 128 %.0f us\n",
               run<0, 1>(threads, out, iters), run<1, 1>(threads, out, iters), run<2, 1>(threads, out, iters), run<3, 1>(threads, out, iters));
    }
    return 0;
}
