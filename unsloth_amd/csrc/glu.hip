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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

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

}  // namespace

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
