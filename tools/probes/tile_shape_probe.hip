// How much HBM bandwidth does a block that WALKS ALONG ROWS get, as a function of the contiguous chunk it reads per row?
// (round 5: the fused SwiGLU kernels own 16 rows x all 14336 columns per block and run at 4.6-5.0 TB/s with every fused
// ingredient compiled out, against 6.7 TB/s of the flat streaming kernel -- profiles/r05_glu_xa_knock.jsonl.)
//   h[M, K] = e[M, K] * g[M, K] (bf16), M = 8192, K = 14336. A 256-thread block owns ROWS rows and walks the columns in steps of
//   4096 / ROWS bytes per row (one 16-byte vector per thread per tensor per step), the next step's loads issued before the
//   current step's arithmetic. ROWS = 0: the flat kernel (every block one contiguous 4 KB piece, no walk).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_shape_probe.hip -o tools/probes/build/tile_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ u32x4 mulv(u32x4 a, u32x4 b) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = __uint_as_float(a[i] << 16), a1 = __uint_as_float(a[i] & 0xffff0000u);
        const float b0 = __uint_as_float(b[i] << 16), b1 = __uint_as_float(b[i] & 0xffff0000u);
        r[i] = (__float_as_uint(a0 * b0) >> 16) | (__float_as_uint(a1 * b1) & 0xffff0000u);
    }
    return r;
}

template <int ROWS, int DEPTH>
__global__ void __launch_bounds__(256) walk_kernel(const unsigned short* __restrict__ E, const unsigned short* __restrict__ G,
                                                   unsigned short* __restrict__ H, int M, int K) {
    constexpr int CHUNK = 2048 / ROWS;                       // elements per row per step
    const int tid = threadIdx.x;
    const int row = blockIdx.x * ROWS + tid / (CHUNK / 8);
    const int col = (tid % (CHUNK / 8)) * 8;
    const size_t base = (size_t)row * K + col;
    const int nsteps = K / CHUNK;
    u32x4 e[DEPTH + 1], g[DEPTH + 1];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        e[d] = __builtin_nontemporal_load((const u32x4*)(E + base + (size_t)d * CHUNK));
        g[d] = __builtin_nontemporal_load((const u32x4*)(G + base + (size_t)d * CHUNK));
    }
    for (int s = 0; s < nsteps; s += DEPTH + 1) {
#pragma unroll
        for (int u = 0; u <= DEPTH; ++u) {
            const int cur = u, nxt = (u + DEPTH) % (DEPTH + 1);
            const int sn = s + u + DEPTH < nsteps ? s + u + DEPTH : nsteps - 1;
            e[nxt] = __builtin_nontemporal_load((const u32x4*)(E + base + (size_t)sn * CHUNK));
            g[nxt] = __builtin_nontemporal_load((const u32x4*)(G + base + (size_t)sn * CHUNK));
            if (s + u < nsteps)
                __builtin_nontemporal_store(mulv(e[cur], g[cur]), (u32x4*)(H + base + (size_t)(s + u) * CHUNK));
        }
    }
}

__global__ void __launch_bounds__(256) flat_kernel(const unsigned short* __restrict__ E, const unsigned short* __restrict__ G,
                                                   unsigned short* __restrict__ H, size_t nvec) {
    const size_t i0 = (size_t)blockIdx.x * 512 + threadIdx.x, i1 = i0 + 256;
    if (i1 < nvec) {
        u32x4 e0 = __builtin_nontemporal_load((const u32x4*)E + i0), g0 = __builtin_nontemporal_load((const u32x4*)G + i0);
        u32x4 e1 = __builtin_nontemporal_load((const u32x4*)E + i1), g1 = __builtin_nontemporal_load((const u32x4*)G + i1);
        __builtin_nontemporal_store(mulv(e0, g0), (u32x4*)H + i0);
        __builtin_nontemporal_store(mulv(e1, g1), (u32x4*)H + i1);
    }
}

// walk_kernel with the columns of a row group SPLIT over `split` adjacent blocks (blockIdx = row group * split + part): each
// block walks K / split columns of its ROWS rows -- how few column parts per row group already restore the sweep?
template <int ROWS>
__global__ void __launch_bounds__(256) walk_split_kernel(const unsigned short* __restrict__ E, const unsigned short* __restrict__ G,
                                                         unsigned short* __restrict__ H, int M, int K, int split) {
    constexpr int CHUNK = 2048 / ROWS;
    const int tid = threadIdx.x;
    const int rg = blockIdx.x / split, part = blockIdx.x % split;
    const int row = rg * ROWS + tid / (CHUNK / 8);
    const int kpart = K / split;
    const size_t base = (size_t)row * K + (size_t)part * kpart + (tid % (CHUNK / 8)) * 8;
    const int nsteps = kpart / CHUNK;
    u32x4 e[2], g[2];
    e[0] = __builtin_nontemporal_load((const u32x4*)(E + base));
    g[0] = __builtin_nontemporal_load((const u32x4*)(G + base));
    for (int s = 0; s < nsteps; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int sn = s + u + 1 < nsteps ? s + u + 1 : nsteps - 1;
            e[u ^ 1] = __builtin_nontemporal_load((const u32x4*)(E + base + (size_t)sn * CHUNK));
            g[u ^ 1] = __builtin_nontemporal_load((const u32x4*)(G + base + (size_t)sn * CHUNK));
            if (s + u < nsteps)
                __builtin_nontemporal_store(mulv(e[u], g[u]), (u32x4*)(H + base + (size_t)(s + u) * CHUNK));
        }
    }
}

// one [ROWS x 4096 / ROWS bytes] tile per tensor per block (TWO adjacent column tiles when TWO), blocks numbered row-group-major:
// the blocks in flight at any moment cover a contiguous window of whole rows, like the flat kernel's sweep
template <int ROWS, bool TWO>
__global__ void __launch_bounds__(256) tileflat_kernel(const unsigned short* __restrict__ E, const unsigned short* __restrict__ G,
                                                       unsigned short* __restrict__ H, int M, int K) {
    constexpr int CHUNK = 2048 / ROWS;
    const int tiles_per_row = K / CHUNK / (TWO ? 2 : 1);
    const int rg = blockIdx.x / tiles_per_row, ct = blockIdx.x % tiles_per_row;
    const int tid = threadIdx.x;
    const int row = rg * ROWS + tid / (CHUNK / 8);
    const size_t base = (size_t)row * K + (size_t)ct * CHUNK * (TWO ? 2 : 1) + (tid % (CHUNK / 8)) * 8;
    u32x4 e0 = __builtin_nontemporal_load((const u32x4*)(E + base)), g0 = __builtin_nontemporal_load((const u32x4*)(G + base));
    if (TWO) {
        u32x4 e1 = __builtin_nontemporal_load((const u32x4*)(E + base + CHUNK)), g1 = __builtin_nontemporal_load((const u32x4*)(G + base + CHUNK));
        __builtin_nontemporal_store(mulv(e1, g1), (u32x4*)(H + base + CHUNK));
    }
    __builtin_nontemporal_store(mulv(e0, g0), (u32x4*)(H + base));
}

template <typename F>
float time_us(F launch, int iters = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms * 1e3f / iters < best) best = ms * 1e3f / iters;
    }
    return best;
}

int main() {
    const int M = 8192, K = 14336;
    const size_t n = (size_t)M * K;
    unsigned short *E, *G, *H;
    hipMalloc(&E, n * 2);
    hipMalloc(&G, n * 2);
    hipMalloc(&H, n * 2);
    std::vector<unsigned short> host(n);
    for (size_t i = 0; i < n; ++i) host[i] = (unsigned short)(0x3f80 + (i * 2654435761u >> 28));
    hipMemcpy(E, host.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(G, host.data(), n * 2, hipMemcpyHostToDevice);
    const double bytes = 3.0 * n * 2;
    auto report = [&](const char* name, float us) {
        printf("{\"kernel\": \"%s\", \"us\": %.1f, \"TBps\": %.2f}\n", name, us, bytes / us / 1e6);
        fflush(stdout);
    };
    report("flat_2x4KB_per_block", time_us([&] { hipLaunchKernelGGL(flat_kernel, dim3((unsigned)((n / 8 + 511) / 512)), dim3(256), 0, 0, E, G, H, n / 8); }));
#define RUN(R, D) report("walk_rows" #R "_chunk" "_depth" #D, time_us([&] { hipLaunchKernelGGL((walk_kernel<R, D>), dim3(M / R), dim3(256), 0, 0, E, G, H, M, K); }))
    RUN(16, 1); RUN(8, 1); RUN(4, 1); RUN(2, 1); RUN(1, 1);
    RUN(16, 2); RUN(8, 2); RUN(4, 2); RUN(2, 2); RUN(1, 2);
    RUN(16, 3); RUN(4, 3); RUN(1, 3);
#define RUNS(R, S) report("walk_rows" #R "_split" #S, time_us([&] { hipLaunchKernelGGL((walk_split_kernel<R>), dim3((M / R) * S), dim3(256), 0, 0, E, G, H, M, K, S); }))
    RUNS(16, 1); RUNS(16, 2); RUNS(16, 4); RUNS(16, 7); RUNS(16, 8); RUNS(16, 14); RUNS(16, 28); RUNS(16, 56);
    RUNS(8, 2); RUNS(8, 4); RUNS(8, 7); RUNS(8, 14); RUNS(8, 28);
#define RUNT(R, TWO_) report("tileflat_rows" #R "_two" #TWO_, time_us([&] { hipLaunchKernelGGL((tileflat_kernel<R, TWO_>), dim3((M / R) * (K / (2048 / R)) / (TWO_ ? 2 : 1)), dim3(256), 0, 0, E, G, H, M, K); }))
    RUNT(16, false); RUNT(16, true); RUNT(8, false); RUNT(8, true); RUNT(4, true); RUNT(1, true);
    return 0;
}
