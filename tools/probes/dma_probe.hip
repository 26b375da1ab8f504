// L2-resident operand-delivery rate per CU on gfx950: LDS-DMA (global_load_lds_dwordx4) vs register loads
// (global_load_dwordx4), GEMM-tile access pattern ([16 rows x 64 B] pieces at an 8 KiB row stride) vs full 128-B
// lines ([8 rows x 128 B]) vs contiguous 1 KiB. One 512-thread block per CU, every block streams the same
// 512-row x 4 KiB panel (2 MiB, L2-resident) tile by tile like a K loop.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) unsigned char lds_u8;

__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
}

// MODE 0: DMA half-line pieces; 1: DMA full-line pieces; 2: DMA contiguous; 3: register loads half-line; 4: reg full-line
template <int MODE, int PIECES>
__global__ void __launch_bounds__(512) probe(const unsigned char* __restrict__ src, int row_bytes, int n_tiles, int iters,
                                             unsigned long long* cycles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    // a 64-KiB "tile" = 512 rows x 128 B (A 256 rows + B 256 rows of one 64-wide bf16 K step)
    const unsigned char* p[PIECES];
#pragma unroll
    for (int c = 0; c < PIECES; ++c) {
        const int piece = c * 8 + wave;                     // 0..63
        if (MODE == 0 || MODE == 3) {                       // [16 rows x 64 B]: piece -> (row group = piece>>1, half = piece&1)
            const int row = (piece >> 1) * 16 + (lane >> 2);
            p[c] = src + (size_t)row * row_bytes + (piece & 1) * 64 + (lane & 3) * 16;
        } else if (MODE == 1 || MODE == 4) {                // [8 rows x 128 B]
            const int row = piece * 8 + (lane >> 3);
            p[c] = src + (size_t)row * row_bytes + (lane & 7) * 16;
        } else {                                            // contiguous KiB
            p[c] = src + (size_t)piece * 1024 + lane * 16;
        }
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < n_tiles; ++t) {
            const size_t koff = (MODE == 2) ? (size_t)t * 65536 : (size_t)t * 128;
            if (MODE <= 2) {
#pragma unroll
                for (int c = 0; c < PIECES; ++c) dma16(p[c] + koff, lds_base + ((t & 1) * 64 + c * 8 + wave) * 1024);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");      // previous tile landed
            } else {
                uint4 v[PIECES];
#pragma unroll
                for (int c = 0; c < PIECES; ++c) v[c] = *reinterpret_cast<const uint4*>(p[c] + koff);
#pragma unroll
                for (int c = 0; c < PIECES; ++c) { acc.x ^= v[c].x; acc.y ^= v[c].y; acc.z ^= v[c].z; acc.w ^= v[c].w; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w ^ smem[lane];
}

template <int MODE>
void run(const char* name, const unsigned char* src, int row_bytes, int n_tiles, unsigned long long* cyc, unsigned* sink, int blocks) {
    const int iters = 20;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE, 8><<<blocks, 512, 131072>>>(src, row_bytes, n_tiles, 2, cyc, sink);
    hipEventRecord(a);
    probe<MODE, 8><<<blocks, 512, 131072>>>(src, row_bytes, n_tiles, iters, cyc, sink);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    const double bytes = (double)iters * n_tiles * 65536.0;
    printf("%-34s blocks %4d  %7.1f B/clk/CU (s_memtime)  wall %.3f ms -> %6.2f TB/s aggregate, eff clock %.2f GHz\n", name, blocks,
           bytes / mean, ms, bytes * blocks / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
}

int main() {
    const int row_bytes = 4096, rows = 512, n_tiles = row_bytes / 128;
    unsigned char* src; unsigned long long* cyc; unsigned* sink;
    hipMalloc(&src, (size_t)rows * row_bytes + 65536 * 64); hipMemset(src, 1, (size_t)rows * row_bytes + 65536 * 64);
    hipMalloc(&cyc, 8 * 1024); hipMalloc(&sink, 64);
    for (int blocks : {256, 32, 8}) {
        run<0>("DMA  [16 rows x 64 B] pieces", src, row_bytes, n_tiles, cyc, sink, blocks);
        run<1>("DMA  [8 rows x 128 B] pieces", src, row_bytes, n_tiles, cyc, sink, blocks);
        run<2>("DMA  contiguous KiB pieces", src, row_bytes, n_tiles, cyc, sink, blocks);
        run<3>("regs [16 rows x 64 B]", src, row_bytes, n_tiles, cyc, sink, blocks);
        run<4>("regs [8 rows x 128 B]", src, row_bytes, n_tiles, cyc, sink, blocks);
    }
    return 0;
}
