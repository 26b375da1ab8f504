// LayerNorm forward / backward for gfx950 (the vision towers' norm: Qwen2-VL ViT blocks, patch merger).
//
// Replaces the Triton kernels of the reference:
//   unsloth/kernels/layernorm.py:25-65   layernorm_forward   mean, XX = x - mean, r = rsqrt(mean(XX^2) + eps),
//                                                            y = (XX r) W + b, everything in fp32, ONE rounding to Y's dtype;
//                                                            r and mean are kept for the backward (:60-61)
//   unsloth/kernels/layernorm.py:68-104  layernorm_backward  normed = (x - mean) r, g = dY W,
//                                                            dX = (g - mean(g) - normed mean(g normed)) r, written OVER dY (:104)
// (no dW / db: the reference returns None for them, Fast_Layernorm.backward :163 -- the norms stay frozen)
//
// HBM-bound. One 256-thread block per row, the row in registers as 16-byte vectors (single HBM read), two block
// reductions through LDS (the variance is computed from the centred values, like the reference, not as E[x^2] - mean^2).
// Rows that are not 16-byte aligned / wider than 8 vectors per thread take the scalar loop.
#include "common.h"

namespace {

constexpr int LN_MAX_ITERS = 8;          // 256 threads x 8 vectors x 8 elements = 16384 columns (bf16)

template <typename T, typename WT, int ITERS>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const T* __restrict__ X, const WT* __restrict__ W, const WT* __restrict__ Bv, T* __restrict__ Y,
                     float* __restrict__ R, float* __restrict__ Mu, int n_cols, int64_t xs, int64_t ys, float eps) {
    __shared__ float red[8];
    constexpr int VEC = Vec16<T>::N;
    const int64_t row = blockIdx.x;
    const T* x = X + row * xs;
    Vec16<T> xv[ITERS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = (threadIdx.x + 256 * i) * VEC;
        if (c < n_cols) xv[i] = ld16(x + c);
        else xv[i].raw = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += to_f32(xv[i].e[j]);
    }
    const float mean = block_sum<4>(s, red) / (float)n_cols;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = (threadIdx.x + 256 * i) * VEC;
        if (c < n_cols) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = to_f32(xv[i].e[j]) - mean; v += d * d; }
        }
    }
    const float inv = rsqrtf(block_sum<4>(v, red + 4) / (float)n_cols + eps);
    if (threadIdx.x == 0) { R[row] = inv; Mu[row] = mean; }
    T* y = Y + row * ys;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = (threadIdx.x + 256 * i) * VEC;
        if (c < n_cols) {
            float wf[VEC], bf_[VEC];
            load_w<WT, VEC>(W + c, wf);
            load_w<WT, VEC>(Bv + c, bf_);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.e[j] = from_f32<T>(((to_f32(xv[i].e[j]) - mean) * inv) * wf[j] + bf_[j]);
            st16(y + c, o);
        }
    }
}

template <typename T, typename WT, int ITERS>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(T* dY, const T* __restrict__ X, const WT* __restrict__ W, const float* __restrict__ R,
                     const float* __restrict__ Mu, int n_cols, int64_t dys, int64_t xs) {
    __shared__ float red[8];
    constexpr int VEC = Vec16<T>::N;
    const int64_t row = blockIdx.x;
    T* dy = dY + row * dys;
    const T* x = X + row * xs;
    const float inv = R[row], mean = Mu[row];
    float g[ITERS][VEC], nrm[ITERS][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = (threadIdx.x + 256 * i) * VEC;
        if (c < n_cols) {
            const Vec16<T> dv = ld16(dy + c), xv = ld16(x + c);
            float wf[VEC];
            load_w<WT, VEC>(W + c, wf);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                nrm[i][j] = (to_f32(xv.e[j]) - mean) * inv;
                g[i][j] = to_f32(dv.e[j]) * wf[j];
                s1 += g[i][j];
                s2 += g[i][j] * nrm[i][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { g[i][j] = 0.f; nrm[i][j] = 0.f; }
        }
    }
    const float m1 = block_sum<4>(s1, red) / (float)n_cols;
    const float m2 = block_sum<4>(s2, red + 4) / (float)n_cols;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = (threadIdx.x + 256 * i) * VEC;
        if (c < n_cols) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.e[j] = from_f32<T>((g[i][j] - m1 - nrm[i][j] * m2) * inv);
            st16(dy + c, o);
        }
    }
}

// generic shapes: scalar loops, three passes over the row (L2-resident)
template <typename T, typename WT>
__global__ void __launch_bounds__(256)
layernorm_fwd_generic(const T* __restrict__ X, const WT* __restrict__ W, const WT* __restrict__ Bv, T* __restrict__ Y,
                      float* __restrict__ R, float* __restrict__ Mu, int n_cols, int64_t xs, int64_t ys, float eps) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const T* x = X + row * xs;
    float s = 0.f;
    for (int c = threadIdx.x; c < n_cols; c += 256) s += to_f32(x[c]);
    const float mean = block_sum<4>(s, red) / (float)n_cols;
    float v = 0.f;
    for (int c = threadIdx.x; c < n_cols; c += 256) { const float d = to_f32(x[c]) - mean; v += d * d; }
    const float inv = rsqrtf(block_sum<4>(v, red + 4) / (float)n_cols + eps);
    if (threadIdx.x == 0) { R[row] = inv; Mu[row] = mean; }
    T* y = Y + row * ys;
    for (int c = threadIdx.x; c < n_cols; c += 256)
        y[c] = from_f32<T>(((to_f32(x[c]) - mean) * inv) * to_f32(W[c]) + to_f32(Bv[c]));
}

template <typename T, typename WT>
__global__ void __launch_bounds__(256)
layernorm_bwd_generic(T* dY, const T* __restrict__ X, const WT* __restrict__ W, const float* __restrict__ R,
                      const float* __restrict__ Mu, int n_cols, int64_t dys, int64_t xs) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    T* dy = dY + row * dys;
    const T* x = X + row * xs;
    const float inv = R[row], mean = Mu[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        const float gg = to_f32(dy[c]) * to_f32(W[c]);
        s1 += gg;
        s2 += gg * ((to_f32(x[c]) - mean) * inv);
    }
    const float m1 = block_sum<4>(s1, red) / (float)n_cols;
    const float m2 = block_sum<4>(s2, red + 4) / (float)n_cols;
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        const float gg = to_f32(dy[c]) * to_f32(W[c]);
        dy[c] = from_f32<T>((gg - m1 - ((to_f32(x[c]) - mean) * inv) * m2) * inv);
    }
}

template <typename T, typename WT>
int ln_fwd(const void* X, const void* W, const void* B, void* Y, float* r, float* mu, int64_t n_rows, int n_cols,
           int64_t xs, int64_t ys, float eps, hipStream_t st) {
    constexpr int VEC = Vec16<T>::N;
    const int iters = (n_cols + 256 * VEC - 1) / (256 * VEC);
    const bool fast = iters <= LN_MAX_ITERS && n_cols % VEC == 0 && xs % VEC == 0 && ys % VEC == 0 && aligned16(X) &&
                      aligned16(Y) && aligned16(W) && aligned16(B);
    dim3 grid((unsigned)n_rows), block(256);
#define LN_F(I) hipLaunchKernelGGL((layernorm_fwd_kernel<T, WT, I>), grid, block, 0, st, (const T*)X, (const WT*)W, \
                                   (const WT*)B, (T*)Y, r, mu, n_cols, xs, ys, eps)
    if (!fast) hipLaunchKernelGGL((layernorm_fwd_generic<T, WT>), grid, block, 0, st, (const T*)X, (const WT*)W,
                                  (const WT*)B, (T*)Y, r, mu, n_cols, xs, ys, eps);
    else if (iters <= 1) LN_F(1);
    else if (iters <= 2) LN_F(2);
    else if (iters <= 4) LN_F(4);
    else LN_F(8);
#undef LN_F
    return uamd_launch_status();
}

template <typename T, typename WT>
int ln_bwd(void* dY, const void* X, const void* W, const float* r, const float* mu, int64_t n_rows, int n_cols,
           int64_t dys, int64_t xs, hipStream_t st) {
    constexpr int VEC = Vec16<T>::N;
    const int iters = (n_cols + 256 * VEC - 1) / (256 * VEC);
    const bool fast = iters <= 4 && n_cols % VEC == 0 && xs % VEC == 0 && dys % VEC == 0 && aligned16(X) &&
                      aligned16(dY) && aligned16(W);
    dim3 grid((unsigned)n_rows), block(256);
#define LN_B(I) hipLaunchKernelGGL((layernorm_bwd_kernel<T, WT, I>), grid, block, 0, st, (T*)dY, (const T*)X, \
                                   (const WT*)W, r, mu, n_cols, dys, xs)
    if (!fast) hipLaunchKernelGGL((layernorm_bwd_generic<T, WT>), grid, block, 0, st, (T*)dY, (const T*)X, (const WT*)W,
                                  r, mu, n_cols, dys, xs);
    else if (iters <= 1) LN_B(1);
    else if (iters <= 2) LN_B(2);
    else LN_B(4);
#undef LN_B
    return uamd_launch_status();
}

}  // namespace

#define LN_DISPATCH(xd, wd, CALL)                                                             \
    if (xd == UAMD_BF16 && wd == UAMD_BF16) { using T = bf16_t; using WT = bf16_t; return CALL; } \
    if (xd == UAMD_F16 && wd == UAMD_F16) { using T = f16_t; using WT = f16_t; return CALL; }     \
    if (xd == UAMD_F32 && wd == UAMD_F32) { using T = float; using WT = float; return CALL; }     \
    if (xd == UAMD_BF16 && wd == UAMD_F32) { using T = bf16_t; using WT = float; return CALL; }   \
    if (xd == UAMD_F16 && wd == UAMD_F32) { using T = f16_t; using WT = float; return CALL; }     \
    return UAMD_ERR_DTYPE;

extern "C" int uamd_layernorm_fwd(const void* X, const void* W, const void* B, void* Y, float* r, float* mu,
                                  int64_t n_rows, int n_cols, int64_t x_row_stride, int64_t y_row_stride, float eps,
                                  int x_dtype, int w_dtype, void* stream) {
    if (n_rows < 0 || n_cols <= 0) return UAMD_ERR_ARG;
    if (n_rows == 0) return UAMD_OK;
    if (!X || !W || !B || !Y || !r || !mu) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(x_dtype, w_dtype, (ln_fwd<T, WT>(X, W, B, Y, r, mu, n_rows, n_cols, x_row_stride, y_row_stride, eps, st)))
}

extern "C" int uamd_layernorm_bwd(void* dY, const void* X, const void* W, const float* r, const float* mu, int64_t n_rows,
                                  int n_cols, int64_t dy_row_stride, int64_t x_row_stride, int x_dtype, int w_dtype,
                                  void* stream) {
    if (n_rows < 0 || n_cols <= 0) return UAMD_ERR_ARG;
    if (n_rows == 0) return UAMD_OK;
    if (!dY || !X || !W || !r || !mu) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(x_dtype, w_dtype, (ln_bwd<T, WT>(dY, X, W, r, mu, n_rows, n_cols, dy_row_stride, x_row_stride, st)))
}
