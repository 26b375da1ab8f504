// Cross-entropy forward (loss + logsumexp per row) and in-place backward.
//
// Replaces the Triton kernels of the reference:
//   unsloth/kernels/cross_entropy_loss.py:35-111   _cross_entropy_forward          (V <= 65536)
//   unsloth/kernels/cross_entropy_loss.py:114-199  _chunked_cross_entropy_forward  (V  > 65536,
//        + the host-side torch.logsumexp / masked_fill_ at :366-370)
//   unsloth/kernels/cross_entropy_loss.py:202-285  _cross_entropy_backward (writes over logits)
//
// The reference needs two code paths because a Triton program holds a whole 65536-wide block;
// Llama-3 (V=128256) always takes the chunked path plus a host reduction. Here ONE 256-thread
// block streams a row of any length once with an online (max, sum) pair per lane, then a
// wave64 shuffle + 4-entry LDS combine: no vocab limit, no second launch, logits read once.
// Backward is a pure streaming kernel over (row, 8192-column chunk).
//
// Integer semantics that must stay exact: label == -100 -> loss 0 and zero gradient row;
// the "- 1" lands exactly on column == label.
#include "common.h"

namespace {

template <bool SOFTCAP, bool SCALE>
__device__ __forceinline__ float ce_transform(float x, float softcap, float scale) {
    if (SCALE) x = scale * x;                         // cross_entropy_loss.py:79-80
    if (SOFTCAP) x = softcap * tanhf(x / softcap);    // cross_entropy_loss.py:82-83
    return x;
}

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    const float M = fmaxf(m, m2);
    if (M == -INFINITY) { m = M; s = 0.f; return; }
    s = s * __expf(m - M) + s2 * __expf(m2 - M);
    m = M;
}

template <typename T, bool SOFTCAP, bool SCALE, bool VECTOR>
__global__ void __launch_bounds__(256)
ce_fwd_kernel(const T* __restrict__ logits, int64_t row_stride, float* __restrict__ loss,
              float* __restrict__ lse, const int64_t* __restrict__ labels, int vocab, float softcap,
              float scale) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red_m[4], red_s[4];
    const int64_t row = blockIdx.x;
    const T* x = logits + row * row_stride;
    float m = -INFINITY, s = 0.f;
    if (VECTOR) {
        const int nvec = vocab / VEC;
        auto absorb = [&](const Vec16<T>& v) {
            float t[VEC];
            float lm = -INFINITY;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                t[j] = ce_transform<SOFTCAP, SCALE>(to_f32(v.e[j]), softcap, scale);
                lm = fmaxf(lm, t[j]);
            }
            const float M = fmaxf(m, lm);
            if (M != -INFINITY) {
                float acc = s * __expf(m - M);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc += __expf(t[j] - M);
                s = acc;
                m = M;
            }
        };
        // two vectors per trip, both loads issued before either is consumed (one 16-byte load in flight per thread left
        // the kernel at 66 % of HBM); the vectors are absorbed in the same order as before: identical results
        int i = threadIdx.x;
        for (; i + 256 < nvec; i += 512) {
            const Vec16<T> v0 = ld16(x + (int64_t)i * VEC);
            const Vec16<T> v1 = ld16(x + (int64_t)(i + 256) * VEC);
            absorb(v0);
            absorb(v1);
        }
        if (i < nvec) absorb(ld16(x + (int64_t)i * VEC));
        for (int c = nvec * VEC + threadIdx.x; c < vocab; c += 256) {
            const float t = ce_transform<SOFTCAP, SCALE>(to_f32(x[c]), softcap, scale);
            online_merge(m, s, t, 1.f);
        }
    } else {
        for (int c = threadIdx.x; c < vocab; c += 256) {
            const float t = ce_transform<SOFTCAP, SCALE>(to_f32(x[c]), softcap, scale);
            online_merge(m, s, t, 1.f);
        }
    }
    // wave combine
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red_m[w] = m; red_s[w] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = red_m[0], S = red_s[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) online_merge(M, S, red_m[i], red_s[i]);
        const float l = M + logf(S);                  // c + log(sum(exp(x - c)))
        lse[row] = l;
        const int64_t label = labels[row];
        float out = 0.f;                              // label == -100 -> 0
        if (label != -100 && label >= 0 && label < vocab) {
            const float xl = ce_transform<SOFTCAP, SCALE>(to_f32(x[label]), softcap, scale);
            out = l - xl;
        }
        loss[row] = out;
    }
}

template <typename T, bool SOFTCAP, bool SCALE>
__device__ __forceinline__ T ce_grad(T xin, int col, int64_t label, float l, float dl, float softcap,
                                     float scale) {
    float x = to_f32(xin);
    if (SCALE) x = x * scale;                         // cross_entropy_loss.py:249-251
    float partial = x;
    if (SOFTCAP) { partial = tanhf(x / softcap); x = softcap * partial; }
    float y = __expf(x - l);
    if ((int64_t)col == label) y -= 1.0f;             // exp(x - lse) - 1 on the label column
    if (SCALE) y = y * scale;
    if (SOFTCAP) y = y * (1.0f - partial * partial);
    return from_f32<T>(dl * y);
}

template <typename T, bool SOFTCAP, bool SCALE, bool VECTOR>
__global__ void __launch_bounds__(256)
ce_bwd_kernel(T* logits, int64_t row_stride, const float* __restrict__ dloss, int64_t dloss_stride,
              const float* __restrict__ lse, const int64_t* __restrict__ labels, int vocab,
              float softcap, float scale, int chunk) {
    constexpr int VEC = Vec16<T>::N;
    const int64_t row = blockIdx.x;
    T* x = logits + row * row_stride;
    const int64_t label = labels[row];
    const float dl = (label != -100) ? dloss[row * dloss_stride] : 0.0f;  // :238-241
    const float l = lse[row];
    const int c0 = blockIdx.y * chunk;
    const int c1 = min(c0 + chunk, vocab);
    if (VECTOR) {
        // chunk is a multiple of VEC; the last chunk may end on a ragged tail
        const int cvec_end = c0 + ((c1 - c0) / VEC) * VEC;
        // a block's chunk is 4 vectors per thread (launch_bwd). A full chunk -- every block of a row but the last -- takes
        // the branch-free form: four loads, then four computes + stores. (A load -> compute -> store loop waits for the
        // previous trip's STORE before it can use the next load, vmcnt being in order; per-lane conditions around the
        // loads make hipcc drain vmcnt at every join.)
        if (c0 + 4 * 256 * VEC <= cvec_end) {
            const int cb = c0 + threadIdx.x * VEC;
            Vec16<T> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld16(x + cb + u * 256 * VEC);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + u * 256 * VEC;
                Vec16<T> o;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    o.e[j] = ce_grad<T, SOFTCAP, SCALE>(v[u].e[j], c + j, label, l, dl, softcap, scale);
                st16(x + c, o);
            }
        } else {
            for (int c = c0 + threadIdx.x * VEC; c < cvec_end; c += 256 * VEC) {
                Vec16<T> v = ld16(x + c), o;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    o.e[j] = ce_grad<T, SOFTCAP, SCALE>(v.e[j], c + j, label, l, dl, softcap, scale);
                st16(x + c, o);
            }
        }
        for (int c = cvec_end + threadIdx.x; c < c1; c += 256)
            x[c] = ce_grad<T, SOFTCAP, SCALE>(x[c], c, label, l, dl, softcap, scale);
    } else {
        for (int c = c0 + threadIdx.x; c < c1; c += 256)
            x[c] = ce_grad<T, SOFTCAP, SCALE>(x[c], c, label, l, dl, softcap, scale);
    }
}

template <typename T>
bool rows_vectorizable(const void* p, int64_t row_stride) {
    return aligned16(p) && (row_stride % Vec16<T>::N == 0);
}

template <typename T>
int launch_fwd(const void* logits, int64_t rs, float* loss, float* lse, const int64_t* labels,
               int64_t n_rows, int vocab, float softcap, float scale, hipStream_t st) {
    const bool vec = rows_vectorizable<T>(logits, rs);
    const bool sc = softcap != 0.f, ls = scale != 0.f;
    dim3 grid((unsigned)n_rows), block(256);
#define L(SC, LS, V) hipLaunchKernelGGL((ce_fwd_kernel<T, SC, LS, V>), grid, block, 0, st, (const T*)logits, rs, loss, lse, labels, vocab, softcap, scale)
#define L2(SC, LS) do { if (vec) L(SC, LS, true); else L(SC, LS, false); } while (0)
    if (sc && ls) L2(true, true); else if (sc) L2(true, false); else if (ls) L2(false, true); else L2(false, false);
#undef L2
#undef L
    return uamd_launch_status();
}

template <typename T>
int launch_bwd(void* logits, int64_t rs, const float* dloss, int64_t ds, const float* lse,
               const int64_t* labels, int64_t n_rows, int vocab, float softcap, float scale,
               hipStream_t st) {
    const bool vec = rows_vectorizable<T>(logits, rs);
    const bool sc = softcap != 0.f, ls = scale != 0.f;
    const int chunk = 256 * Vec16<T>::N * 4;
    dim3 grid((unsigned)n_rows, (unsigned)((vocab + chunk - 1) / chunk)), block(256);
#define L(SC, LS, V) hipLaunchKernelGGL((ce_bwd_kernel<T, SC, LS, V>), grid, block, 0, st, (T*)logits, rs, dloss, ds, lse, labels, vocab, softcap, scale, chunk)
#define L2(SC, LS) do { if (vec) L(SC, LS, true); else L(SC, LS, false); } while (0)
    if (sc && ls) L2(true, true); else if (sc) L2(true, false); else if (ls) L2(false, true); else L2(false, false);
#undef L2
#undef L
    return uamd_launch_status();
}

}  // namespace

// loss[row] = logsumexp(row) - x[label] (0 when label == -100); lse[row] = logsumexp(row).
// softcap / logit_scale == 0 disable the respective transform (reference convention).
extern "C" int uamd_cross_entropy_forward(const void* logits, int64_t logits_row_stride, float* loss,
                                          float* logsumexp, const int64_t* labels, int64_t n_rows,
                                          int vocab_size, float logit_softcapping,
                                          float logit_scaling, int dtype, void* stream) {
    if (n_rows < 0 || vocab_size <= 0 || n_rows > 0x7fffffffLL) return UAMD_ERR_ARG;
    if (n_rows == 0) return UAMD_OK;
    UAMD_DISPATCH_FLOAT(dtype, return (launch_fwd<T>(logits, logits_row_stride, loss, logsumexp, labels,
                                                     n_rows, vocab_size, logit_softcapping,
                                                     logit_scaling, (hipStream_t)stream)))
    return UAMD_ERR_DTYPE;
}

// logits <- dloss[row] * d(loss)/d(logits), in place (same dtype as logits).
extern "C" int uamd_cross_entropy_backward(void* logits, int64_t logits_row_stride,
                                           const float* dloss, int64_t dloss_stride,
                                           const float* logsumexp, const int64_t* labels,
                                           int64_t n_rows, int vocab_size, float logit_softcapping,
                                           float logit_scaling, int dtype, void* stream) {
    if (n_rows < 0 || vocab_size <= 0 || n_rows > 0x7fffffffLL) return UAMD_ERR_ARG;
    if (n_rows == 0) return UAMD_OK;
    UAMD_DISPATCH_FLOAT(dtype, return (launch_bwd<T>(logits, logits_row_stride, dloss, dloss_stride,
                                                     logsumexp, labels, n_rows, vocab_size,
                                                     logit_softcapping, logit_scaling,
                                                     (hipStream_t)stream)))
    return UAMD_ERR_DTYPE;
}
