// Gated-MLP activations: SwiGLU, GeGLU (exact erf), GeGLU (tanh approx); forward and the
// fused in-place backward.
//
// Replaces the Triton kernels of the reference:
//   unsloth/kernels/swiglu.py:27-47    _fg_kernel              h = (e*sigmoid(e)).to(dtype) * g
//   unsloth/kernels/swiglu.py:67-109   _DWf_DW_dfg_kernel      DW<-h, e<-df, g<-de  (in place)
//   unsloth/kernels/geglu.py:31-53     _exact_forward_kernel
//   unsloth/kernels/geglu.py:74-123    _exact_backward_kernel
//   unsloth/kernels/geglu.py:142-167   _approx_forward_kernel
//   unsloth/kernels/geglu.py:188-244   _approx_backward_kernel
//
// HBM-bound streaming kernels: 16-byte vectors per lane, two vectors per thread with every load of both issued
// before the first use, one pass per 256-thread block over an UNCAPPED grid (the hardware dispatcher streams the
// blocks; the round-1 version ran 2048 persistent blocks with a grid-stride loop: 5.85 -> 6.7 TB/s forward,
// 5.04 -> 5.66 TB/s backward at 8192 tokens, profiles/r02_hbm_ab.jsonl; the reference's Triton kernels on the same
// box: 5.5 / 6.0 TB/s, profiles/r02_ref_triton_microbench.jsonl). Inputs are read exactly once (2-3 x 235 MB at
// 8192 tokens: larger than the 256 MiB Infinity Cache) and loaded non-temporally. 64-bit indexing always (the
// reference switches to int64 only above 2^31 elements, swiglu.py:20-24).
//
// Rounding points follow the reference exactly: f is rounded to the activation dtype before
// it is multiplied by g; h/df/dg are products IN the activation dtype; de is computed in
// fp32 from the rounded dg and rounded once.
#include "common.h"

namespace {

enum { ACT_SWIGLU = 0, ACT_GEGLU_EXACT = 1, ACT_GEGLU_APPROX = 2 };
#ifndef UAMD_GLU_KNOCK
#define UAMD_GLU_KNOCK 0      /* != 0: timing-only builds of glu_xa_kernel with one ingredient compiled out (results are wrong) */
#endif

__device__ __forceinline__ float sigmoidf_(float x) { return uamd_sigmoid(x); }      // common.h

// forward activation f(e) in fp32, plus (for backward) df/de
template <int ACT>
__device__ __forceinline__ void act_fwd(float e, float& f) {
    if (ACT == ACT_SWIGLU) {
        f = e * sigmoidf_(e);                                   // swiglu.py:41
    } else if (ACT == ACT_GEGLU_EXACT) {
        f = 0.5f * e * (erff(0.70710678118654752440f * e) + 1.0f);  // geglu.py:47
    } else {
        const float s = 0.7978845608028654f;                    // sqrt(2/pi), geglu.py:156
        f = 0.5f * e * (tanhf(s * e * (1.0f + 0.044715f * e * e)) + 1.0f);
    }
}

template <int ACT>
__device__ __forceinline__ void act_bwd(float e, float& f, float& dfde) {
    if (ACT == ACT_SWIGLU) {
        const float se = sigmoidf_(e);
        f = se * e;                                             // swiglu.py:93
        dfde = se * (1.0f + e * (1.0f - se));                   // swiglu.py:103
    } else if (ACT == ACT_GEGLU_EXACT) {
        const float fp = 0.5f * (erff(0.70710678118654752440f * e) + 1.0f);  // geglu.py:100
        f = fp * e;
        const float t = 0.3989422804014327f;                    // 1/sqrt(2*pi)
        dfde = fp + t * e * __expf(-0.5f * e * e);               // geglu.py:113
    } else {
        const float s = 0.7978845608028654f;
        const float a = s * e;
        const float b = a * 0.044715f * e * e;
        const float T = 1.0f + tanhf(a + b);
        const float T2 = 0.5f * T;
        const float Q2 = -T2 * (T - 2.0f) * (a + 3.0f * b);     // geglu.py:221-225
        dfde = T2 + Q2;
        f = T2 * e;
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256)
glu_fwd_kernel(const T* __restrict__ E, const T* __restrict__ G, T* __restrict__ H, int64_t n, int mode) {
    constexpr int VEC = Vec16<T>::N;
    const int64_t nvec = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        Vec16<T> e = ld16_m(E + i * VEC, mode), g = ld16_m(G + i * VEC, mode), h;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float f;
            act_fwd<ACT>(to_f32(e.e[j]), f);
            h.e[j] = from_f32<T>(round_to<T>(f) * to_f32(g.e[j]));
        }
        st16_m(H + i * VEC, h, mode);
    }
    // tail (n not a multiple of VEC)
    if (blockIdx.x == 0) {
        for (int64_t k = nvec * VEC + threadIdx.x; k < n; k += 256) {
            float f;
            act_fwd<ACT>(to_f32(E[k]), f);
            H[k] = from_f32<T>(round_to<T>(f) * to_f32(G[k]));
        }
    }
}

template <typename T, int ACT>
__device__ __forceinline__ void bwd_one(T dw, T e, T g, T& h, T& df, T& de) {
    float f32, dfde;
    const float ef = to_f32(e);
    act_bwd<ACT>(ef, f32, dfde);
    const float f = round_to<T>(f32);                 // f_row.to(DW_row.dtype)
    const float DW = to_f32(dw), gg = to_f32(g);
    h = from_f32<T>(f * gg);                          // h  = f * g      (in dtype)
    df = from_f32<T>(DW * f);                         // df = DW * f     (in dtype)
    const float dg = round_to<T>(DW * gg);            // dg = DW * g     (in dtype)
    if (ACT == ACT_SWIGLU) {
        // same association as swiglu.py:103: (dg * se) * (1 + e * (1 - se))
        const float se = sigmoidf_(ef);
        de = from_f32<T>(dg * se * (1.0f + ef * (1.0f - se)));
    } else {
        de = from_f32<T>(dg * dfde);                  // de = dg.float() * df/de, rounded once
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) glu_bwd_kernel(T* DW, T* E, T* G, int64_t n, int mode) {
    constexpr int VEC = Vec16<T>::N;
    const int64_t nvec = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        Vec16<T> dw = ld16_m(DW + i * VEC, mode), e = ld16_m(E + i * VEC, mode), g = ld16_m(G + i * VEC, mode);
        Vec16<T> h, df, de;
#pragma unroll
        for (int j = 0; j < VEC; ++j) bwd_one<T, ACT>(dw.e[j], e.e[j], g.e[j], h.e[j], df.e[j], de.e[j]);
        st16_m(DW + i * VEC, h, mode);   // swiglu.py:107-109
        st16_m(E + i * VEC, df, mode);
        st16_m(G + i * VEC, de, mode);
    }
    if (blockIdx.x == 0) {
        for (int64_t k = nvec * VEC + threadIdx.x; k < n; k += 256) {
            T h, df, de;
            bwd_one<T, ACT>(DW[k], E[k], G[k], h, df, de);
            DW[k] = h; E[k] = df; G[k] = de;
        }
    }
}

// Two vectors per thread, both tensors' loads of both vectors in flight before the first use; one pass per block
// (no grid-stride loop: the hardware dispatcher streams 256-thread blocks, which is how the reference's Triton
// kernel reaches 6.0 TB/s on this shape).
template <typename T, int ACT>
__global__ void __launch_bounds__(256) glu_bwd2_kernel(T* DW, T* E, T* G, int64_t n, int mode) {
    constexpr int VEC = Vec16<T>::N;
    const int64_t nvec = n / VEC;
    const int64_t i0 = (int64_t)blockIdx.x * 512 + threadIdx.x, i1 = i0 + 256;
    if (i1 < nvec) {
        Vec16<T> dw0 = ld16_m(DW + i0 * VEC, mode), e0 = ld16_m(E + i0 * VEC, mode), g0 = ld16_m(G + i0 * VEC, mode);
        Vec16<T> dw1 = ld16_m(DW + i1 * VEC, mode), e1 = ld16_m(E + i1 * VEC, mode), g1 = ld16_m(G + i1 * VEC, mode);
        Vec16<T> h, df, de;
#pragma unroll
        for (int j = 0; j < VEC; ++j) bwd_one<T, ACT>(dw0.e[j], e0.e[j], g0.e[j], h.e[j], df.e[j], de.e[j]);
        st16_m(DW + i0 * VEC, h, mode);
        st16_m(E + i0 * VEC, df, mode);
        st16_m(G + i0 * VEC, de, mode);
#pragma unroll
        for (int j = 0; j < VEC; ++j) bwd_one<T, ACT>(dw1.e[j], e1.e[j], g1.e[j], h.e[j], df.e[j], de.e[j]);
        st16_m(DW + i1 * VEC, h, mode);
        st16_m(E + i1 * VEC, df, mode);
        st16_m(G + i1 * VEC, de, mode);
    } else if (i0 < nvec) {
        Vec16<T> dw = ld16_m(DW + i0 * VEC, mode), e = ld16_m(E + i0 * VEC, mode), g = ld16_m(G + i0 * VEC, mode);
        Vec16<T> h, df, de;
#pragma unroll
        for (int j = 0; j < VEC; ++j) bwd_one<T, ACT>(dw.e[j], e.e[j], g.e[j], h.e[j], df.e[j], de.e[j]);
        st16_m(DW + i0 * VEC, h, mode);
        st16_m(E + i0 * VEC, df, mode);
        st16_m(G + i0 * VEC, de, mode);
    }
    if (blockIdx.x == 0) {
        for (int64_t k = nvec * VEC + threadIdx.x; k < n; k += 256) {
            T h, df, de;
            bwd_one<T, ACT>(DW[k], E[k], G[k], h, df, de);
            DW[k] = h; E[k] = df; G[k] = de;
        }
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256)
glu_fwd2_kernel(const T* __restrict__ E, const T* __restrict__ G, T* __restrict__ H, int64_t n, int mode) {
    constexpr int VEC = Vec16<T>::N;
    const int64_t nvec = n / VEC;
    const int64_t i0 = (int64_t)blockIdx.x * 512 + threadIdx.x, i1 = i0 + 256;
    auto one = [&](const Vec16<T>& e, const Vec16<T>& g, int64_t i) {
        Vec16<T> h;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float f;
            act_fwd<ACT>(to_f32(e.e[j]), f);
            h.e[j] = from_f32<T>(round_to<T>(f) * to_f32(g.e[j]));
        }
        st16_m(H + i * VEC, h, mode);
    };
    if (i1 < nvec) {
        Vec16<T> e0 = ld16_m(E + i0 * VEC, mode), g0 = ld16_m(G + i0 * VEC, mode);
        Vec16<T> e1 = ld16_m(E + i1 * VEC, mode), g1 = ld16_m(G + i1 * VEC, mode);
        one(e0, g0, i0);
        one(e1, g1, i1);
    } else if (i0 < nvec) {
        Vec16<T> e0 = ld16_m(E + i0 * VEC, mode), g0 = ld16_m(G + i0 * VEC, mode);
        one(e0, g0, i0);
    }
    if (blockIdx.x == 0) {
        for (int64_t k = nvec * VEC + threadIdx.x; k < n; k += 256) {
            float f;
            act_fwd<ACT>(to_f32(E[k]), f);
            H[k] = from_f32<T>(round_to<T>(f) * to_f32(G[k]));
        }
    }
}

// UAMD_TUNE_GLU_VAR: 0 = 2048-block grid-stride kernels, 1 = the same kernels with one vector per thread and an
// uncapped grid, 2 = two vectors per thread, uncapped grid
inline unsigned grid_for(int64_t nvec, int var) {
    int64_t blocks = (nvec + 255) / 256;
    if (blocks < 1) blocks = 1;
    const int64_t cap = var == 0 ? 256 * 8 : 0x7fffffffLL;  // 256 CUs x 8 resident 256-thread blocks
    return (unsigned)(blocks < cap ? blocks : cap);
}

template <typename T, int ACT>
int launch_fwd(const void* e, const void* g, void* h, int64_t n, hipStream_t st) {
    if (!aligned16(e) || !aligned16(g) || !aligned16(h)) return UAMD_ERR_ALIGN;
    const int var = uamd_tuning_get(UAMD_TUNE_GLU_VAR);
    const int64_t nvec = n / Vec16<T>::N;
    if (var == 2 && (nvec + 511) / 512 < 0x7fffffffLL)
        hipLaunchKernelGGL((glu_fwd2_kernel<T, ACT>), dim3((unsigned)((nvec + 511) / 512 > 0 ? (nvec + 511) / 512 : 1)),
                           dim3(256), 0, st, (const T*)e, (const T*)g, (T*)h, n, uamd_tuning_get(UAMD_TUNE_STREAM_NT) ^ 1);
    else
        hipLaunchKernelGGL((glu_fwd_kernel<T, ACT>), dim3(grid_for(nvec, var)), dim3(256), 0, st,
                           (const T*)e, (const T*)g, (T*)h, n, uamd_tuning_get(UAMD_TUNE_STREAM_NT) ^ 1);
    return uamd_launch_status();
}
template <typename T, int ACT>
int launch_bwd(void* dw, void* e, void* g, int64_t n, hipStream_t st) {
    if (!aligned16(dw) || !aligned16(e) || !aligned16(g)) return UAMD_ERR_ALIGN;
    const int var = uamd_tuning_get(UAMD_TUNE_GLU_VAR);
    const int64_t nvec = n / Vec16<T>::N;
    if (var == 2 && (nvec + 511) / 512 < 0x7fffffffLL)
        hipLaunchKernelGGL((glu_bwd2_kernel<T, ACT>), dim3((unsigned)((nvec + 511) / 512 > 0 ? (nvec + 511) / 512 : 1)),
                           dim3(256), 0, st, (T*)dw, (T*)e, (T*)g, n, uamd_tuning_get(UAMD_TUNE_STREAM_NT) ^ 1);
    else
        hipLaunchKernelGGL((glu_bwd_kernel<T, ACT>), dim3(grid_for(nvec, var)), dim3(256), 0, st,
                           (T*)dw, (T*)e, (T*)g, n, uamd_tuning_get(UAMD_TUNE_STREAM_NT) ^ 1);
    return uamd_launch_status();
}

template <int ACT>
int fwd(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream) {
    if (n < 0) return UAMD_ERR_ARG;
    if (n == 0) return UAMD_OK;
    UAMD_DISPATCH_FLOAT(dtype, return (launch_fwd<T, ACT>(e, g, h, n, (hipStream_t)stream)))
    return UAMD_ERR_DTYPE;
}
template <int ACT>
int bwd(void* dw, void* e, void* g, int64_t n, int dtype, void* stream) {
    if (n < 0) return UAMD_ERR_ARG;
    if (n == 0) return UAMD_OK;
    UAMD_DISPATCH_FLOAT(dtype, return (launch_bwd<T, ACT>(dw, e, g, n, (hipStream_t)stream)))
    return UAMD_ERR_DTYPE;
}

// ---------------------------------------------------------------------------------------------------------------------
// The gated activation FUSED with the skinny LoRA products that would otherwise re-read its output (fast_lora.py:93-96:
// `h = f(e) * g` then `h @ A_down^T`; :157, :172-189: `h, df, de` then `df @ B_up`, `de @ B_gate`): the activation kernel
// already has every element of h / df / de in registers, and [T, 14336] is the widest activation of the layer (235 MB at
// 8192 tokens) -- reading it again for a rank-16 product costs as much as the product's whole launch
// (profiles/r03d_step_sequence.csv: lora_xa2 55 us forward, 64 + 69 us backward per layer, next to 105 / 216 us of the
// activation kernels themselves).
// The 4 partial [16 x R] tiles of a block are summed through LDS in fixed order; outputs: fp32 [M, ld_out] (columns R..
// zero-filled up to out_cols) and, optionally, the same sums rounded to the activation dtype into the GEMM's rank-block
// operand (uamd_gemm_group.lora_xk) at its column offset. (First version, measured +2.6 % on the step: lanes read 64 bytes
// per row straight into the MFMA operand layout -- half lines, nothing prefetched; profiles/r03h_glu_fused_ab.txt.)
typedef __attribute__((ext_vector_type(8))) __bf16 glu_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 glu_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float glu_f32x4_t;
template <typename T> struct GluMfma;
template <> struct GluMfma<bf16_t> {
    typedef glu_bf16x8_t frag;
    static __device__ __forceinline__ glu_f32x4_t run(frag a, frag b, glu_f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct GluMfma<f16_t> {
    typedef glu_f16x8_t frag;
    static __device__ __forceinline__ glu_f32x4_t run(frag a, frag b, glu_f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

struct GluXaOut {
    float* out;          // fp32 [M, ld_out]: columns [0, R) = the product, [R, out_cols) = 0
    int64_t ld_out;
    int R, out_cols;
    void* out_k;         // activation dtype [M, ld_k] or null: columns [0, R) = rounded product, [R, k_cols) = 0
    int64_t ld_k;
    int k_cols;
    const void* W;       // [R, K] activation dtype, K contiguous
    int64_t ldw;
};

// NS = number of products (1: forward, h x A_down; 2: backward, df x B_up^T and de x B_gate^T), NT = 16-rank tiles each.
// Block = 256 threads = 4 waves, 16 rows x all K in tiles of 256 columns (three blocks per CU):
//   * global access is streaming: a wave instruction reads / writes two 512-byte row pieces (thread t, vector v: row
//     8 v + 2 wave + (lane >> 5), columns 8 (lane & 31) ..), the next tile's loads are issued before the current tile is
//     computed;
//   * the 16-bit results also go into an LDS tile [16 rows][256 + 8 columns] (row stride 528 B: the 16 rows of a
//     fragment read start 4 banks apart), double-buffered, ONE barrier per tile;
//   * wave w then contracts k-steps w and w + 4 of the tile on the matrix cores: A operand = ds_read_b128 of
//     the tile (row l15, 8 columns at 32 ks + 8 l4), B operand = 16 bytes of the LoRA factor's row nt * 16 + l15 (L2),
//     requested at the top of the tile so that the activation arithmetic hides the latency.
constexpr int GX_TK = 256;                       // columns per tile
constexpr int GX_LD = GX_TK * 2 + 16;            // LDS row stride in bytes
constexpr int GX_TILE = 16 * GX_LD;              // 8,448 B per operand tile

// KS = 1: the block above (4 waves, two vectors per thread per tensor, two k-steps per wave). KS = 2 (round 5): the SAME
// 16-row x 256-column tile worked by 8 waves -- one vector per thread per tensor, one k-step per wave: half the streamed
// registers and fragments per thread (the backward instance went from 176 to <= 128 VGPRs), so two 8-wave blocks fit a CU
// and every SIMD has 4 waves to hide the activation arithmetic behind (the kernel is VALU-heavy: ~26 instructions per
// element forward, two of them quarter-rate, at 2 waves per SIMD: profiles/r05_glu_xa_ab.jsonl).
template <typename T, int ACT, int NS, int NT, int KS, int PD>
__global__ void __launch_bounds__(256 * KS, 2 * KS)
glu_xa_kernel(T* __restrict__ DW, T* __restrict__ E, T* __restrict__ G, T* __restrict__ H, int M, int K, int64_t ld,
              GluXaOut o0, GluXaOut o1, float* __restrict__ part, int* __restrict__ counters, int nparts, int tpb) {
    typedef typename GluMfma<T>::frag frag_t;
    constexpr int NWAVE = 4 * KS, NTHR = 256 * KS;
    constexpr int NV = 2 / KS, NQ = 2 / KS;                              // vectors per thread per tensor, k-steps per wave
    constexpr int RED = NWAVE * 16 * NS * NT * 16 * 4;                   // bytes of the final reduction buffer
    constexpr int LDS_B = 2 * NS * GX_TILE > RED ? 2 * NS * GX_TILE : RED;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // nparts > 1 (round 5): the columns of a 16-row group are SPLIT over nparts adjacent workgroups of tpb tiles each
    // (blockIdx = row group * nparts + part) -- see the note on the grid's memory sweep in front of launch_xa
    const int rg = nparts > 1 ? (int)(blockIdx.x / (unsigned)nparts) : (int)blockIdx.x;
    const int cpart = nparts > 1 ? (int)(blockIdx.x - (unsigned)rg * (unsigned)nparts) : 0;
    const int m0 = rg * 16;
    glu_f32x4_t acc[NS][NT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = glu_f32x4_t{0.f, 0.f, 0.f, 0.f};
    const T* W0 = (const T*)o0.W;
    const T* W1 = (const T*)o1.W;
    int64_t woff0[NT], woff1[NT];                                       // rank rows past R re-read the last valid row
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        woff0[t] = (int64_t)min(t * 16 + l15, o0.R - 1) * o0.ldw + l4 * 8;
        woff1[t] = NS > 1 ? (int64_t)min(t * 16 + l15, o1.R - 1) * o1.ldw + l4 * 8 : 0;
    }
    // streaming side: vector v of this thread = row (16 / NV) v + 2 wave + (lane >> 5) of the block, columns 8 (lane & 31) ..
    const int srow = 2 * wave + (lane >> 5), scol = (lane & 31) * 8;
    int64_t roff[NV];
    bool row_ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int r = m0 + (16 / NV) * v + srow;
        row_ok[v] = r < M;
        roff[v] = (int64_t)(row_ok[v] ? r : M - 1) * ld;                 // clamped rows are computed and never stored
    }
    const int ntiles_all = (K + GX_TK - 1) / GX_TK;
    const int it_begin = nparts > 1 ? cpart * tpb : 0;
    const int ntiles = nparts > 1 ? min(it_begin + tpb, ntiles_all) : ntiles_all;       // (exclusive END of this block's tiles)
    // two register sets for the streamed operands AND the LoRA factor's fragments, used alternately (the loop is unrolled
    // by two: a `cur = next` copy at the end of a tile would make the compiler wait for the prefetch right there). One set
    // = everything tile `it` consumes, requested one tile ahead in consumption order (vmcnt retires in order: data first,
    // fragments second, then the previous tile's stores -- no wait ever covers a younger request).
    struct Regs {
        Vec16<T> e[NV], g[NV], dw[NV];
        uint4 wf0[NQ][NT], wf1[NS > 1 ? NQ : 1][NT];
    };
    Regs A, B, C;                                                        // (C: the third set of the depth-2 prefetch, PD == 2)
    auto load_data = [&](int it, Regs& r) {
        const int c = it * GX_TK + scol;
        const int cc = c + 8 <= K ? c : 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            r.e[v] = ld16_nt(E + roff[v] + cc);
            r.g[v] = ld16_nt(G + roff[v] + cc);
            if (NS > 1) r.dw[v] = ld16_nt(DW + roff[v] + cc);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_frags = [&](int it, Regs& r) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int kc = it * GX_TK + (wave + NWAVE * q) * 32;
            const int kk = kc + l4 * 8 + 8 <= K ? kc : 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#if UAMD_GLU_KNOCK & 1     /* knock-out build (tools/glu_xa_ab.py): no factor-fragment loads */
                r.wf0[q][t] = make_uint4(kk, kk, kk, kk);
                if (NS > 1) r.wf1[q][t] = make_uint4(kk, kk, kk, kk);
#else
                r.wf0[q][t] = *reinterpret_cast<const uint4*>(W0 + woff0[t] + kk);
                if (NS > 1) r.wf1[q][t] = *reinterpret_cast<const uint4*>(W1 + woff1[t] + kk);
#endif
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_tile = [&](int it, Regs& r) {
        load_data(it, r);
        if constexpr (PD == 1) load_frags(it, r);
    };
    auto tile = [&](int it, Regs& r, Regs& rn, int it_next, auto prefetch) {
        unsigned char* buf = smem + (it & 1) * NS * GX_TILE;
        // PD == 2: the factor fragments of THIS tile (L2-resident: half a microsecond, hidden by the activation arithmetic
        // below) go out BEFORE the data of the tile two steps ahead: vmcnt retires in order, so the wait in front of the MFMAs
        // covers them and everything older, never the younger prefetch. One fragment set instead of three: the registers
        // that keep the backward instance at 4 waves per SIMD.
        if constexpr (PD == 2) load_frags(it, r);
        // compile-time yes / no, never a run-time branch (behind a branch the compiler drains vmcnt at the join)
        if constexpr (decltype(prefetch)::value) load_tile(it_next, rn);
        const int c = it * GX_TK + scol;
        const bool col_ok = c + 8 <= K;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            Vec16<T> v0, v1;
            v1.raw = make_uint4(0, 0, 0, 0);
            if (NS == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f;
                    act_fwd<ACT>(to_f32(r.e[v].e[j]), f);
                    v0.e[j] = from_f32<T>(round_to<T>(f) * to_f32(r.g[v].e[j]));
                }
                if (col_ok && row_ok[v]) st16_nt(H + roff[v] + c, v0);
            } else {
                Vec16<T> h;
#pragma unroll
                for (int j = 0; j < 8; ++j) bwd_one<T, ACT>(r.dw[v].e[j], r.e[v].e[j], r.g[v].e[j], h.e[j], v0.e[j], v1.e[j]);
                if (col_ok && row_ok[v]) {
                    st16_nt(DW + roff[v] + c, h);
                    st16_nt(E + roff[v] + c, v0);
                    st16_nt(G + roff[v] + c, v1);
                }
            }
            if (!col_ok) {                                               // columns past K contribute nothing
                v0.raw = make_uint4(0, 0, 0, 0);
                v1.raw = make_uint4(0, 0, 0, 0);
            }
#if !(UAMD_GLU_KNOCK & 2)
            unsigned char* dst = buf + ((16 / NV) * v + srow) * GX_LD + scol * 2;
            *reinterpret_cast<uint4*>(dst) = v0.raw;
            if (NS > 1) *reinterpret_cast<uint4*>(dst + GX_TILE) = v1.raw;
#endif
        }
#if UAMD_GLU_KNOCK & 2     /* knock-out build: no LDS hand-off, no barrier, no MFMA -- the activation on this tiling alone */
        (void)buf;
        return;
#endif
        // the tile is complete (and buf ^ 1 is free again). NOT __syncthreads(): that drains vmcnt too, i.e. waits for
        // the NEXT tile's loads issued above -- the prefetch would buy nothing (first build: 4.1 / 4.65 TB/s)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const unsigned char* src = buf + l15 * GX_LD + ((wave + NWAVE * q) * 32 + l4 * 8) * 2;
            union { uint4 r; frag_t f; } a0, a1, w;
            a0.r = *reinterpret_cast<const uint4*>(src);
            if (NS > 1) a1.r = *reinterpret_cast<const uint4*>(src + GX_TILE);
            const bool k_ok = it * GX_TK + (wave + NWAVE * q) * 32 + l4 * 8 + 8 <= K;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w.r = k_ok ? r.wf0[q][t] : make_uint4(0, 0, 0, 0);
                acc[0][t] = GluMfma<T>::run(a0.f, w.f, acc[0][t]);
                if (NS > 1) {
                    w.r = k_ok ? r.wf1[q][t] : make_uint4(0, 0, 0, 0);
                    acc[1][t] = GluMfma<T>::run(a1.f, w.f, acc[1][t]);
                }
            }
        }
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    load_tile(it_begin, A);
    if constexpr (PD == 1) {
        // pairs of tiles in a branch-free body; an odd last tile after the loop (a conditional second tile inside the loop
        // is a join at which hipcc drains vmcnt -- prefetched loads AND the previous tile's stores -- every iteration). The
        // last tile is simply fetched once more.
        int it = it_begin;
        for (; it + 1 < ntiles; it += 2) {
            tile(it, A, B, it + 1 < ntiles ? it + 1 : it, yes);
            tile(it + 1, B, A, it + 2 < ntiles ? it + 2 : it + 1, yes);
        }
        if (it < ntiles) tile(it, A, B, it, yes);
    } else {
        // PD == 2: every tile is requested TWO tile steps before it is consumed (three register sets in rotation, the loop
        // unrolled by three): the kernel's rate is bytes in flight / latency, and one tile ahead is 24-49 KB per CU
        // (profiles/r05_glu_xa_ab.jsonl: neither cheaper arithmetic nor twice the waves moved it).
        load_tile(it_begin + 1 < ntiles ? it_begin + 1 : it_begin, B);
        int it = it_begin;
        for (; it + 4 < ntiles; it += 3) {
            tile(it, A, C, it + 2, yes);
            tile(it + 1, B, A, it + 3, yes);
            tile(it + 2, C, B, it + 4, yes);
        }
        // 1 .. 4 tiles left (A holds tile `it`, B tile it + 1); what is still missing is fetched, nothing twice
        const int rem = ntiles - it;
        if (rem >= 3) {
            tile(it, A, C, it + 2, yes);
            if (rem == 4) tile(it + 1, B, A, it + 3, yes);
            else tile(it + 1, B, A, 0, no);
            tile(it + 2, C, B, 0, no);
            if (rem == 4) tile(it + 3, A, B, 0, no);
        } else {
            tile(it, A, B, 0, no);
            if (rem == 2) tile(it + 1, B, A, 0, no);
        }
    }
    // ---- NWAVE partial tiles -> LDS -> fixed-order sum. acc[s][t][i] = C[row 4 l4 + i][rank t * 16 + l15]
    constexpr int RW = NS * NT * 16;                                     // floats per row of the reduction buffer
    __syncthreads();                                                     // every wave is done with the operand tiles
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wave * 16 + 4 * l4 + i) * RW + (s * NT + t) * 16 + l15] = acc[s][t][i];
    __syncthreads();
    if (nparts > 1) {
        // ---- this block's [16 x RW] sums are a PARTIAL over its columns: to the workspace; the block that finishes the row group
        //      LAST (an arrival counter per row group, zero on entry, left zero) adds the parts in PART ORDER -- a fixed order
        //      whoever arrives last: deterministic -- and writes `out` / `out_k`.
        //      No __threadfence() / release-acquire atomics: at agent scope they are `buffer_wbl2`, a write-back of the whole L2,
        //      per workgroup while 0.7-1.4 GB of results stream through it (a one-tile-per-workgroup build of this scheme: 4.7 ms
        //      instead of 0.27). The partials are agent-scope atomic stores (write-through) and agent-scope atomic loads (never
        //      served from a non-coherent line): waiting for this thread's stores to be acknowledged (vmcnt(0)) before the
        //      barrier in front of the counter's increment is all the ordering there is to establish.
        __shared__ int s_last;
        float* mine = part + (int64_t)blockIdx.x * (16 * RW);
        for (int idx = tid; idx < 16 * RW; idx += NTHR) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) v += red[w * 16 * RW + idx];
            __hip_atomic_store(mine + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) s_last = __hip_atomic_fetch_add(counters + rg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nparts - 1;
        __syncthreads();
        if (!s_last) return;
        float* pg = part + (int64_t)rg * nparts * (16 * RW);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const GluXaOut& o = s == 0 ? o0 : o1;
            const int wide = o.out_cols > o.k_cols ? o.out_cols : o.k_cols;
            for (int idx = tid; idx < 16 * wide; idx += NTHR) {
                const int mm = idx / wide, cl = idx - mm * wide;
                if (m0 + mm >= M) continue;
                float v = 0.f;
                if (cl < o.R) {
                    float* q = pg + mm * RW + s * NT * 16 + cl;
                    for (int p_ = 0; p_ < nparts; ++p_)
                        v += __hip_atomic_load(q + (int64_t)p_ * (16 * RW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (cl < o.out_cols) o.out[(int64_t)(m0 + mm) * o.ld_out + cl] = v;
                if (o.out_k != nullptr && cl < o.k_cols) ((T*)o.out_k)[(int64_t)(m0 + mm) * o.ld_k + cl] = from_f32<T>(v);
            }
        }
        if (tid == 0) __hip_atomic_store(counters + rg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const GluXaOut& o = s == 0 ? o0 : o1;
        for (int idx = tid; idx < 16 * o.out_cols; idx += NTHR) {
            const int mm = idx / o.out_cols, c = idx - mm * o.out_cols;
            if (m0 + mm >= M) continue;
            float v = 0.f;
            if (c < o.R) {
                const float* q = red + mm * RW + s * NT * 16 + c;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) v += q[w * 16 * RW];
            }
            o.out[(int64_t)(m0 + mm) * o.ld_out + c] = v;
        }
        if (o.out_k != nullptr) {
            for (int idx = tid; idx < 16 * o.k_cols; idx += NTHR) {
                const int mm = idx / o.k_cols, c = idx - mm * o.k_cols;
                if (m0 + mm >= M) continue;
                float v = 0.f;
                if (c < o.R) {
                    const float* q = red + mm * RW + s * NT * 16 + c;
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) v += q[w * 16 * RW];
                }
                ((T*)o.out_k)[(int64_t)(m0 + mm) * o.ld_k + c] = from_f32<T>(v);
            }
        }
    }
}

// The grid's memory sweep (round 5). What held the fused kernels at 4.6-5.0 TB/s was neither their arithmetic (a cheaper
// sigmoid: nothing), nor occupancy (8 waves per block: +1 %), nor prefetch depth (two tiles ahead: nothing), nor anything the
// fusion adds (LDS hand-off, barrier, MFMAs and factor loads compiled out: 141 of 150 us remain, profiles/r05_glu_xa_knock.jsonl)
// -- it is the ORDER in which the grid touches memory. 512 workgroups that each walk along their own 16 rows sweep the matrix
// column slab by column slab: every DRAM page is visited dozens of times, 512 bytes at a time. tools/probes/tile_shape_probe.hip
// isolates it on e * g alone (profiles/r05_tile_shape_probe.jsonl): workgroups walking 16 rows 4.4-4.8 TB/s at any prefetch
// depth; the same walk with the columns of a row group split over 2 / 4 / 14 ADJACENT workgroups 5.1 / 5.3 / 5.8; one tile per
// workgroup numbered row-group-major 6.4; the flat kernel 6.6. One tile per workgroup was built and measured: 267 us against
// 148 -- 28,672 workgroups each paying a prologue, three barriers, a store acknowledgement and an atomic round trip for 8 KB
// per tensor -- and parts of 4 tiles lose to the unsplit walk as well (fill and drain of the prefetch per 4 tiles). What ships keeps
// the pipelined walk and splits the columns of a row group over TWO adjacent workgroups where that measured faster (the rule in
// launch_xa): blockIdx = row group * parts + part, the rank products of a part to a workspace ([row group][part][16][NS * NT *
// 16] fp32: well under 1 % of the kernel's traffic), summed in part order by whichever workgroup finishes the row group last.
// The probe measured e * g at 28,672-byte rows; at other widths the unsplit walk is not as far from the flat sweep and the split
// buys nothing (launch_xa), so "the order of the sweep" is the mechanism for Llama-3-8B's MLP width, not a law.
// is the current device one whose ISA the fence-free split hand-off was written for (see launch_xa)?
inline bool split_handoff_ok() {
    static int ok[64] = {0};            // 0 = unknown, 1 = yes, 2 = no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (ok[dev] == 0) {
        hipDeviceProp_t pr;
        bool y = false;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) {
            const char* a = pr.gcnArchName;
            auto starts = [&](const char* q) { for (int i = 0; q[i]; ++i) if (a[i] != q[i]) return false; return true; };
            y = starts("gfx942") || starts("gfx950");
        }
        ok[dev] = y ? 1 : 2;
    }
    return ok[dev] == 1;
}

constexpr int GX_TPB = 4;     // (workspace sizing: parts of at least this many tiles)

template <typename T, int ACT, int NS>
int launch_xa(void* dw, void* e, void* g, void* h, int M, int K, int64_t ld, const GluXaOut& o0, const GluXaOut& o1,
              hipStream_t st, float* ws, int* counters) {
    const int R = NS > 1 ? (o0.R > o1.R ? o0.R : o1.R) : o0.R;
    // UAMD_TUNE_GLU_XA: 3 = 2 + the column split (needs the workspace); 2 = 8 waves per 16-row block, tiles requested two steps
    // ahead; 1 = 8 waves, one step ahead; 0 = 4 waves, one step ahead (rounds 3-4)
    // (SwiGLU only -- the one activation the callers fuse, kernels/utils.py glu_fwd_xa: the GeGLU instances keep the round-3 form.
    // The BACKWARD at ranks above 16 too: two products x two rank tiles of fragments do not fit the 128 registers of 4 waves per
    // SIMD -- hipcc's bf16 build answers with a full vmcnt(0) in the tile loop, tests/test_isa_loop_waits.py -- so it stays at
    // 4 waves x 176 registers)
    const int xv = ACT == ACT_SWIGLU ? uamd_tuning_get(UAMD_TUNE_GLU_XA) : 0;
    const int ntiles = (K + GX_TK - 1) / GX_TK;
    // When to split, and into how many parts: measured, not derived (profiles/r05_glu_xa_ab.jsonl: parts of 4 / 8 / 14 / 28 tiles
    // and two / seven EVEN parts at [8192 | 4096 | 2048] x 14336, 4096 x 18944, 8192 x 11008, [8192 | 2048] x 5632, 16384 x
    // 14336). TWO EVEN PARTS win wherever a row is a whole number of 4 KB pages -- Llama-3-8B / Mistral-7B's 14336 columns: every
    // workgroup of the unsplit walk then starts its rows at the same offset inside a page -- at every height (forward 149 -> 135,
    // 78 -> 70, 57 -> 39 us; backward 311 -> 282, 160 -> 146, 104 -> 76 us at 8192 / 4096 / 2048 rows), and for short grids
    // (<= 2048 rows: 128 workgroups for 256 CUs) at any width (2048 x 5632 backward 43 -> 35 us). Elsewhere the split is neutral
    // (4096 x 18944: 92 / 191 vs 94 / 192 us) or a few percent slower (8192 x 11008, 8192 x 5632), and uneven or many parts lose
    // outright (three parts of 28 + 28 + 18 tiles at 18944 columns: +20 %; parts of 4 tiles: the prefetch's fill and drain per 4
    // tiles of work). UAMD_TUNE_GLU_XA: 3 = this rule, 8 = two even parts always (A/B).
    // The split's hand-off (relaxed agent-scope stores + s_waitcnt vmcnt(0) + barrier + relaxed counter increment, no fence) is
    // outside the C++ memory model: it is correct because gfx942 / gfx950 write such stores through and count them in vmcnt. Any
    // other target gets the unsplit kernel. `counters` must be all zero on entry: a launch that died mid-flight leaves them
    // dirty -- the host wrapper (kernels/utils.py) re-zeroes its counters whenever a launch of this family returned an error.
    const bool kSplitOk = split_handoff_ok();
    const bool two = kSplitOk && ntiles >= 8 && (xv == 8 || ((int64_t)K * (int64_t)sizeof(T)) % 4096 == 0 || M <= 2048);
    const int tpb = two ? (ntiles + 1) / 2 : ntiles;
    int nparts = (xv >= 3 && ws != nullptr && counters != nullptr) ? (ntiles + tpb - 1) / tpb : 1;
    if (nparts < 2) nparts = 1;
    const int64_t blocks = (int64_t)((M + 15) / 16) * nparts;
    if (blocks > 0x7fffffffLL) return UAMD_ERR_ARG;
    const dim3 grid((unsigned)blocks);
#define GLU_XA_ARGS (T*)dw, (T*)e, (T*)g, (T*)h, M, K, ld, o0, o1, ws, counters, nparts, tpb
#define GLU_XA_LAUNCH(NT_)                                                                                                   \
    do {                                                                                                                     \
        if constexpr (ACT != ACT_SWIGLU || (NS > 1 && NT_ > 1)) hipLaunchKernelGGL((glu_xa_kernel<T, ACT, NS, NT_, 1, 1>), grid, dim3(256), 0, st, GLU_XA_ARGS); \
        else if (xv >= 2) hipLaunchKernelGGL((glu_xa_kernel<T, ACT, NS, NT_, 2, 2>), grid, dim3(512), 0, st, GLU_XA_ARGS);   \
        else if (xv == 1) hipLaunchKernelGGL((glu_xa_kernel<T, ACT, NS, NT_, 2, 1>), grid, dim3(512), 0, st, GLU_XA_ARGS);   \
        else hipLaunchKernelGGL((glu_xa_kernel<T, ACT, NS, NT_, 1, 1>), grid, dim3(256), 0, st, GLU_XA_ARGS);                \
    } while (0)
    if (R <= 16) GLU_XA_LAUNCH(1);
    else if (R <= 32) GLU_XA_LAUNCH(2);
    else GLU_XA_LAUNCH(4);
#undef GLU_XA_LAUNCH
#undef GLU_XA_ARGS
    return uamd_launch_status();
}

int check_xa_out(const GluXaOut& o, int K) {
    if (!o.out || !o.W || o.R < 1 || o.R > 64 || o.out_cols < o.R) return UAMD_ERR_ARG;
    if ((o.ldw & 7) || !aligned16(o.W)) return UAMD_ERR_ALIGN;
    if (o.out_k && o.k_cols < o.R) return UAMD_ERR_ARG;
    (void)K;
    return UAMD_OK;
}

// floats of workspace the column split needs for an [M, K] launch (uamd_glu_xa_workspace)
int64_t xa_ws_floats(int M, int K, int NS, int R) {
    const int NT = R <= 16 ? 1 : (R <= 32 ? 2 : 4);
    const int ntiles = (K + GX_TK - 1) / GX_TK;
    return (int64_t)((M + 15) / 16) * ((ntiles + GX_TPB - 1) / GX_TPB) * 16 * (NS * NT * 16);
}

template <int NS>
int glu_xa_entry(int act, void* dw, void* e, void* g, void* h, int M, int K, int64_t ld, const GluXaOut& o0,
                 const GluXaOut& o1, int dtype, void* stream, float* ws = nullptr, int64_t ws_floats = 0, int* counters = nullptr) {
    if (M < 0 || K <= 0 || !e || !g || (NS == 1 ? !h : !dw)) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 7) || (ld & 7) || !aligned16(e) || !aligned16(g) || (NS == 1 ? !aligned16(h) : !aligned16(dw))) return UAMD_ERR_ALIGN;
    int rc = check_xa_out(o0, K);
    if (rc) return rc;
    if (NS > 1 && (rc = check_xa_out(o1, K))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (ws != nullptr && ws_floats < xa_ws_floats(M, K, NS, NS > 1 ? (o0.R > o1.R ? o0.R : o1.R) : o0.R)) return UAMD_ERR_ARG;
#define GLU_XA_CASE(TT, DC)                                                                          \
    if (dtype == DC) {                                                                              \
        if (act == ACT_SWIGLU) return launch_xa<TT, ACT_SWIGLU, NS>(dw, e, g, h, M, K, ld, o0, o1, st, ws, counters);          \
        if (act == ACT_GEGLU_EXACT) return launch_xa<TT, ACT_GEGLU_EXACT, NS>(dw, e, g, h, M, K, ld, o0, o1, st, ws, counters); \
        if (act == ACT_GEGLU_APPROX) return launch_xa<TT, ACT_GEGLU_APPROX, NS>(dw, e, g, h, M, K, ld, o0, o1, st, ws, counters); \
        return UAMD_ERR_ARG;                                                                        \
    }
    GLU_XA_CASE(bf16_t, UAMD_BF16)
    GLU_XA_CASE(f16_t, UAMD_F16)
#undef GLU_XA_CASE
    return UAMD_ERR_DTYPE;
}

}  // namespace

// h[M, K] = act(e) * g  AND  XA[M, R] = h @ W^T (W = the down projection's LoRA A, [R, K]) in one pass over e and g.
// act: 0 = SwiGLU, 1 = GeGLU exact, 2 = GeGLU tanh. e / g / h share the row stride `ld` (elements). out: fp32 [M, ld_out],
// columns [R, out_cols) zero-filled; out_k (may be null): the sums rounded to `dtype`, columns [R, k_cols) zero-filled.
extern "C" int uamd_glu_fwd_xa(int act, const void* e, const void* g, void* h, int M, int K, int64_t ld, const void* W,
                               int64_t ldw, int R, float* out, int64_t ld_out, int out_cols, void* out_k, int64_t ld_k,
                               int k_cols, int dtype, void* stream) {
    GluXaOut o0{out, ld_out, R, out_cols, out_k, ld_k, k_cols, W, ldw};
    return glu_xa_entry<1>(act, nullptr, const_cast<void*>(e), const_cast<void*>(g), h, M, K, ld, o0, o0, dtype, stream);
}

// The in-place backward (DW <- h, e <- df, g <- de) AND P_u[M, R_u] = df @ Wu^T, P_g[M, R_g] = de @ Wg^T (Wu = B_up^T,
// Wg = B_gate^T, [R, K] K-contiguous) in one pass. Outputs as above, one set per product.
extern "C" int uamd_glu_bwd_xa(int act, void* DW, void* e, void* g, int M, int K, int64_t ld,
                               const void* Wu, int64_t ldwu, int Ru, float* out_u, int64_t ld_out_u, int out_cols_u,
                               void* out_k_u, int64_t ld_k_u, int k_cols_u,
                               const void* Wg, int64_t ldwg, int Rg, float* out_g, int64_t ld_out_g, int out_cols_g,
                               void* out_k_g, int64_t ld_k_g, int k_cols_g, int dtype, void* stream) {
    GluXaOut o0{out_u, ld_out_u, Ru, out_cols_u, out_k_u, ld_k_u, k_cols_u, Wu, ldwu};
    GluXaOut o1{out_g, ld_out_g, Rg, out_cols_g, out_k_g, ld_k_g, k_cols_g, Wg, ldwg};
    return glu_xa_entry<2>(act, DW, e, g, nullptr, M, K, ld, o0, o1, dtype, stream);
}

// The same two calls with a workspace: `ws` = at least uamd_glu_xa_workspace(M, K, n_products, max rank) floats, `counters` =
// (M + 15) / 16 ints that are ZERO on entry (the kernel leaves them zero): the column split (the note in front of launch_xa). One
// workspace per device and stream: two launches in flight on different streams must not share it.
extern "C" int64_t uamd_glu_xa_workspace(int M, int K, int n_products, int max_rank) {
    if (M < 0 || K <= 0 || n_products < 1 || n_products > 2 || max_rank < 1 || max_rank > 64) return -1;
    return xa_ws_floats(M, K, n_products, max_rank);
}
extern "C" int uamd_glu_fwd_xa_ws(int act, const void* e, const void* g, void* h, int M, int K, int64_t ld, const void* W,
                                  int64_t ldw, int R, float* out, int64_t ld_out, int out_cols, void* out_k, int64_t ld_k,
                                  int k_cols, float* ws, int64_t ws_floats, int* counters, int dtype, void* stream) {
    GluXaOut o0{out, ld_out, R, out_cols, out_k, ld_k, k_cols, W, ldw};
    return glu_xa_entry<1>(act, nullptr, const_cast<void*>(e), const_cast<void*>(g), h, M, K, ld, o0, o0, dtype, stream, ws,
                           ws_floats, counters);
}
extern "C" int uamd_glu_bwd_xa_ws(int act, void* DW, void* e, void* g, int M, int K, int64_t ld,
                                  const void* Wu, int64_t ldwu, int Ru, float* out_u, int64_t ld_out_u, int out_cols_u,
                                  void* out_k_u, int64_t ld_k_u, int k_cols_u,
                                  const void* Wg, int64_t ldwg, int Rg, float* out_g, int64_t ld_out_g, int out_cols_g,
                                  void* out_k_g, int64_t ld_k_g, int k_cols_g, float* ws, int64_t ws_floats, int* counters,
                                  int dtype, void* stream) {
    GluXaOut o0{out_u, ld_out_u, Ru, out_cols_u, out_k_u, ld_k_u, k_cols_u, Wu, ldwu};
    GluXaOut o1{out_g, ld_out_g, Rg, out_cols_g, out_k_g, ld_k_g, k_cols_g, Wg, ldwg};
    return glu_xa_entry<2>(act, DW, e, g, nullptr, M, K, ld, o0, o1, dtype, stream, ws, ws_floats, counters);
}

// ---------------------------------------------------------------------------------------------------------------------
// QuickGELU, y = x * sigmoid(1.702 x): the activation of Qwen2-VL's vision MLP (fc1 -> act -> fc2; BASELINE config 4). The
// reference leaves it to the zoo compiler's fused graph (unsloth/models/vision.py:881-1990); here one streaming kernel each way,
// fp32 arithmetic, one rounding: forward y from x; backward dx = dy * (s + 1.702 x s (1 - s)), s = sigmoid(1.702 x), written IN
// PLACE over dy (x is kept: fc1's output is what autograd saved anyway).
namespace {
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) quick_gelu_kernel(const T* __restrict__ X, T* __restrict__ Y, int64_t n) {
    constexpr int VEC = Vec16<T>::N;
    const int64_t nvec = n / VEC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        Vec16<T> x = ld16_m(X + i * VEC, 0), y;
        if (BWD) y = ld16_m(Y + i * VEC, 0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float v = to_f32(x.e[j]);
            const float sg = uamd_sigmoid(1.702f * v);
            if (BWD) y.e[j] = from_f32<T>(to_f32(y.e[j]) * (sg + 1.702f * v * sg * (1.0f - sg)));
            else y.e[j] = from_f32<T>(v * sg);
        }
        st16_m(Y + i * VEC, y, 0);
    }
    if (blockIdx.x == 0) {
        for (int64_t k = nvec * VEC + threadIdx.x; k < n; k += 256) {
            const float v = to_f32(X[k]);
            const float sg = uamd_sigmoid(1.702f * v);
            Y[k] = from_f32<T>(BWD ? to_f32(Y[k]) * (sg + 1.702f * v * sg * (1.0f - sg)) : v * sg);
        }
    }
}
template <typename T, bool BWD>
int launch_quick_gelu(const void* x, void* y, int64_t n, hipStream_t st) {
    if (!aligned16(x) || !aligned16(y)) return UAMD_ERR_ALIGN;
    const int64_t nvec = n / Vec16<T>::N;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((quick_gelu_kernel<T, BWD>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)y, n);
    return uamd_launch_status();
}
}  // namespace

extern "C" int uamd_quick_gelu_forward(const void* x, void* y, int64_t n, int dtype, void* stream) {
    if (n < 0 || (n > 0 && (!x || !y))) return UAMD_ERR_ARG;
    if (n == 0) return UAMD_OK;
    UAMD_DISPATCH_FLOAT(dtype, return (launch_quick_gelu<T, false>(x, y, n, (hipStream_t)stream)))
    return UAMD_ERR_DTYPE;
}
extern "C" int uamd_quick_gelu_backward(const void* x, void* dy_dx, int64_t n, int dtype, void* stream) {
    if (n < 0 || (n > 0 && (!x || !dy_dx))) return UAMD_ERR_ARG;
    if (n == 0) return UAMD_OK;
    UAMD_DISPATCH_FLOAT(dtype, return (launch_quick_gelu<T, true>(x, dy_dx, n, (hipStream_t)stream)))
    return UAMD_ERR_DTYPE;
}

extern "C" int uamd_swiglu_fg(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream) {
    return fwd<ACT_SWIGLU>(e, g, h, n, dtype, stream);
}
extern "C" int uamd_swiglu_DWf_DW_dfg(void* DW, void* e, void* g, int64_t n, int dtype, void* stream) {
    return bwd<ACT_SWIGLU>(DW, e, g, n, dtype, stream);
}
extern "C" int uamd_geglu_exact_forward(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream) {
    return fwd<ACT_GEGLU_EXACT>(e, g, h, n, dtype, stream);
}
extern "C" int uamd_geglu_exact_backward(void* DW, void* e, void* g, int64_t n, int dtype, void* stream) {
    return bwd<ACT_GEGLU_EXACT>(DW, e, g, n, dtype, stream);
}
extern "C" int uamd_geglu_approx_forward(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream) {
    return fwd<ACT_GEGLU_APPROX>(e, g, h, n, dtype, stream);
}
extern "C" int uamd_geglu_approx_backward(void* DW, void* e, void* g, int64_t n, int dtype, void* stream) {
    return bwd<ACT_GEGLU_APPROX>(DW, e, g, n, dtype, stream);
}
