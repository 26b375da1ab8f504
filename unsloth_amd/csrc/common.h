// Common device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave size is 64 on CDNA; every reduction below is written for 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "unsloth_amd.h"   // C ABI: error codes, dtype codes, entry-point prototypes

__attribute__((visibility("hidden"))) int uamd_tuning_get(int knob);   // abi.hip; library-internal (not part of the C ABI)

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

#define UAMD_WAVE 64

// 16-byte vector of T (8 x 16-bit or 4 x 32-bit)
template <typename T> struct Vec16 {
    static constexpr int N = 16 / sizeof(T);
    union {
        uint4 raw;
        T e[16 / sizeof(T)];
    };
    __device__ __forceinline__ Vec16() {}
};

template <typename T>
__device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> v;
    v.raw = *reinterpret_cast<const uint4*>(p);
    return v;
}
template <typename T>
__device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
    *reinterpret_cast<uint4*>(p) = v.raw;
}

// streaming (non-temporal) variants: data read or written exactly once by the whole grid
typedef __attribute__((ext_vector_type(4))) unsigned int uamd_u32x4;
template <typename T>
__device__ __forceinline__ Vec16<T> ld16_nt(const T* p) {
    Vec16<T> v;
    const uamd_u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const uamd_u32x4*>(p));
    v.raw = make_uint4(r[0], r[1], r[2], r[3]);
    return v;
}
template <typename T>
__device__ __forceinline__ void st16_nt(T* p, const Vec16<T>& v) {
    const uamd_u32x4 r = {v.raw.x, v.raw.y, v.raw.z, v.raw.w};
    __builtin_nontemporal_store(r, reinterpret_cast<uamd_u32x4*>(p));
}
// mode bit0: non-temporal loads, bit1: non-temporal stores (UAMD_TUNE_STREAM_NT)
template <typename T>
__device__ __forceinline__ Vec16<T> ld16_m(const T* p, int mode) { return (mode & 1) ? ld16_nt(p) : ld16(p); }
template <typename T>
__device__ __forceinline__ void st16_m(T* p, const Vec16<T>& v, int mode) {
    if (mode & 2) st16_nt(p, v); else st16(p, v);
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// round a float through T (the "rounding point" of the reference kernels)
template <typename T> __device__ __forceinline__ float round_to(float x) { return (float)((T)x); }

// sigmoid(x) = 1 / (1 + exp(-x)) with the IEEE division (11 of the ~26 instructions the SwiGLU forward spends per element).
// Round 5 measured the cheaper v_rcp_f32 + one Newton step (two fmas, <= 1 fp32 ulp off) in its place: the fused activation
// kernels were NOT faster forward and 7 % slower backward (profiles/r05_glu_xa_ab.jsonl, "default" = Newton against
// "libunsloth_amd_ieee.so") -- they are bound by bytes in flight, not by VALU issue -- so the exact quotient stays.
// One definition for glu.hip and the decode GEMV's SwiGLU prologue (bit-identical to each other, tests/test_gpu_decode.py).
__device__ __forceinline__ float uamd_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of NW waves; `red` is NW floats of LDS. Result is broadcast.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (NW == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    if (NW == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// load VEC consecutive elements of WT (vector loads of 8/16/32 bytes) as floats
template <typename WT, int VEC>
__device__ __forceinline__ void load_w(const WT* __restrict__ w, float* out) {
    constexpr int BYTES = VEC * (int)sizeof(WT);
    static_assert(BYTES == 8 || BYTES == 16 || BYTES == 32, "unsupported vector width");
    if constexpr (BYTES == 8) {
        union { uint2 raw; WT e[VEC]; } v;
        v.raw = *reinterpret_cast<const uint2*>(w);
#pragma unroll
        for (int j = 0; j < VEC; ++j) out[j] = to_f32(v.e[j]);
    } else {
        constexpr int WN = Vec16<WT>::N;
#pragma unroll
        for (int k = 0; k < VEC / WN; ++k) {
            Vec16<WT> v = ld16(w + k * WN);
#pragma unroll
            for (int j = 0; j < WN; ++j) out[k * WN + j] = to_f32(v.e[j]);
        }
    }
}


// GEMM epilogue shared by the MFMA kernels (gemm.hip, gemm256.hip): a lane holds C[m][n .. n+3] in fp32; `bias` (the
// base layer's bias, activation dtype, NULL = none) is added BEFORE the single rounding to the activation dtype and
// `accumulate` adds the existing C (dX += ...).
template <typename T, bool ACC, bool BIAS>
__device__ __forceinline__ void store_c4(T* dst, float v0, float v1, float v2, float v3, int n, int N, bool vec_ok,
                                         const T* bias) {
    // ACC / BIAS are COMPILE-TIME: with run-time flags every one of a thread's 32 stores carried its own scalar loads of
    // the kernel arguments + branches, and the 256-tile GEMM lost 12 % at K = 4096 (profiles/r03l_gemm_epilogue_ab.jsonl)
    float v[4] = {v0, v1, v2, v3};
    if (n + 3 < N && vec_ok) {
        union { uint2 raw; T e[4]; } o;
        if (BIAS) {
            union { uint2 raw; T e[4]; } bv;
            if ((reinterpret_cast<uintptr_t>(bias + n) & 7) == 0) {
                bv.raw = *reinterpret_cast<const uint2*>(bias + n);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv.e[r] = bias[n + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f32(bv.e[r]);
        }
        if (ACC) {
            o.raw = *reinterpret_cast<const uint2*>(dst);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f32(o.e[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o.e[r] = from_f32<T>(v[r]);
        *reinterpret_cast<uint2*>(dst) = o.raw;
    } else {
        for (int r = 0; r < 4 && n + r < N; ++r) {
            float x = v[r];
            if (BIAS) x += to_f32(bias[n + r]);
            if (ACC) x += to_f32(dst[r]);
            dst[r] = from_f32<T>(x);
        }
    }
}

// run `epi(acc_c, bias_c)` (two std::integral_constant<bool, ..> tags) for the launch's (accumulate, bias) combination:
// ONE pair of uniform branches per tile instead of one per store
#define UAMD_EPILOGUE_DISPATCH(epi, accumulate, bias_ptr)                                                        \
    do {                                                                                                         \
        if ((bias_ptr) == nullptr) {                                                                             \
            if (accumulate) epi(std::integral_constant<bool, true>{}, std::integral_constant<bool, false>{});   \
            else epi(std::integral_constant<bool, false>{}, std::integral_constant<bool, false>{});             \
        } else {                                                                                                 \
            if (accumulate) epi(std::integral_constant<bool, true>{}, std::integral_constant<bool, true>{});    \
            else epi(std::integral_constant<bool, false>{}, std::integral_constant<bool, true>{});              \
        }                                                                                                        \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int uamd_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? UAMD_OK : (int)e;
}

// dispatch helper over the three activation dtypes
#define UAMD_DISPATCH_FLOAT(dtype, ...)                       \
    switch (dtype) {                                          \
        case UAMD_F32: { using T = float; __VA_ARGS__; break; }  \
        case UAMD_F16: { using T = f16_t; __VA_ARGS__; break; }  \
        case UAMD_BF16: { using T = bf16_t; __VA_ARGS__; break; } \
        default: return UAMD_ERR_DTYPE;                       \
    }
