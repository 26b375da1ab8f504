// Rotary position embedding (rotate-half form), in place, forward and backward.
//
// Replaces the Triton kernels of the reference:
//   unsloth/kernels/rope_embedding.py:104-166  _rope_embedding     (dense [B*T, H*D], pos = row % seqlen)
//   unsloth/kernels/rope_embedding.py:23-98    _rope_embedding_QK  (Q and K in one launch, strided
//                                              [B,H,T,D] views, optional int32 position gather)
// Both are one kernel here: the dense form is the strided form with
// (batch,head,seq) strides = (seqlen*row_stride, head_dim, row_stride) and no K.
//
// HBM-bound. One 256-thread block per token; a thread owns one 16-byte vector column of the
// half head (its cos/sin vector is loaded once) and walks the heads of Q then K, so the
// table is read once per token instead of once per head.
//
// Arithmetic dtype: fp32 single-rounding when either side is fp32 or the
// dtypes differ; when Q and the table share a 16-bit dtype the reference multiplies and
// subtracts in that dtype (rope_embedding.py:77-89,153-158), i.e. three roundings:
// rn(rn(q0*cos) - rn(q1*sin)). NATIVE reproduces exactly that.
#include "common.h"

namespace {

struct RopeArgs {
    void* Q; int64_t q_bs, q_hs, q_ss;
    void* K; int64_t k_bs, k_hs, k_ss;
    const void* cos; int64_t cos_rs;
    const void* sin; int64_t sin_rs;
    const int32_t* idx;
    int64_t n_rows;  // batch * seqlen
    int seqlen, n_heads_q, n_heads_k, head_dim, backward;
    // multimodal RoPE (Qwen2-VL "mrope"): three position streams (temporal, height, width), pos3 = int32 [3, n_rows];
    // rotary pair j takes its angle from stream 0 for j < sec1, 1 for j < sec2, else 2 (NULL: ordinary RoPE)
    const int32_t* pos3;
    int sec1, sec2;
};

__device__ __forceinline__ int64_t rope_position(const RopeArgs& a, int64_t row, int pair) {
    if (a.pos3) {
        const int stream = pair < a.sec1 ? 0 : (pair < a.sec2 ? 1 : 2);
        return (int64_t)a.pos3[(int64_t)stream * a.n_rows + row];
    }
    return a.idx ? (int64_t)a.idx[row] : (row % a.seqlen);            // rope_embedding.py:46-56
}

template <typename T, bool NATIVE>
__device__ __forceinline__ void rotate(float q0, float q1, float c, float s, T& o0, T& o1) {
    if (NATIVE) {
        const float a = round_to<T>(q0 * c), b = round_to<T>(q1 * s);
        const float d = round_to<T>(q1 * c), e = round_to<T>(q0 * s);
        o0 = from_f32<T>(a - b);
        o1 = from_f32<T>(d + e);
    } else {
        o0 = from_f32<T>(q0 * c - q1 * s);
        o1 = from_f32<T>(q1 * c + q0 * s);
    }
}

template <typename T, typename TT, bool NATIVE>
__global__ void __launch_bounds__(256) rope_vec_kernel(RopeArgs a) {
    constexpr int VEC = Vec16<T>::N;
    const int half = a.head_dim >> 1;
    const int vecs = half / VEC;           // vectors per half head
    const int64_t row = blockIdx.x;
    const int v = threadIdx.x % vecs;
    const int slot = threadIdx.x / vecs;
    const int nslots = 256 / vecs;
    if (slot >= nslots) return;
    const int64_t pos = rope_position(a, row, v * VEC);      // section boundaries are multiples of VEC here
    float c[VEC], s[VEC];
    load_w<TT, VEC>((const TT*)a.cos + pos * a.cos_rs + v * VEC, c);
    load_w<TT, VEC>((const TT*)a.sin + pos * a.sin_rs + v * VEC, s);
    if (a.backward) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = -s[j];
    }
    const int64_t b = row / a.seqlen, t = row - b * a.seqlen;
    const int total = a.n_heads_q + a.n_heads_k;
    for (int h = slot; h < total; h += nslots) {
        T* p = (h < a.n_heads_q)
                   ? (T*)a.Q + b * a.q_bs + (int64_t)h * a.q_hs + t * a.q_ss
                   : (T*)a.K + b * a.k_bs + (int64_t)(h - a.n_heads_q) * a.k_hs + t * a.k_ss;
        Vec16<T> x0 = ld16(p + v * VEC), x1 = ld16(p + half + v * VEC), o0, o1;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            rotate<T, NATIVE>(to_f32(x0.e[j]), to_f32(x1.e[j]), c[j], s[j], o0.e[j], o1.e[j]);
        st16(p + v * VEC, o0);
        st16(p + half + v * VEC, o1);
    }
}

// scalar fallback (odd head_dim/2, unaligned views)
template <typename T, typename TT, bool NATIVE>
__global__ void __launch_bounds__(256) rope_scalar_kernel(RopeArgs a) {
    const int half = a.head_dim >> 1;
    const int64_t row = blockIdx.x;
    const int64_t b = row / a.seqlen, t = row - b * a.seqlen;
    const int total = a.n_heads_q + a.n_heads_k;
    for (int w = threadIdx.x; w < total * half; w += 256) {
        const int h = w / half, j = w - h * half;
        T* p = (h < a.n_heads_q)
                   ? (T*)a.Q + b * a.q_bs + (int64_t)h * a.q_hs + t * a.q_ss
                   : (T*)a.K + b * a.k_bs + (int64_t)(h - a.n_heads_q) * a.k_hs + t * a.k_ss;
        const int64_t pos = rope_position(a, row, j);
        const float c = to_f32(((const TT*)a.cos + pos * a.cos_rs)[j]);
        float s = to_f32(((const TT*)a.sin + pos * a.sin_rs)[j]);
        if (a.backward) s = -s;
        T o0, o1;
        rotate<T, NATIVE>(to_f32(p[j]), to_f32(p[j + half]), c, s, o0, o1);
        p[j] = o0;
        p[j + half] = o1;
    }
}

template <typename T, typename TT, bool NATIVE>
int launch(const RopeArgs& a, hipStream_t st) {
    constexpr int VEC = Vec16<T>::N;
    const int half = a.head_dim / 2;
    bool vec_ok = (half % VEC == 0) && (half / VEC <= 256) && aligned16(a.Q) &&
                  (a.q_bs % VEC == 0) && (a.q_hs % VEC == 0) && (a.q_ss % VEC == 0) &&
                  (a.cos_rs % VEC == 0) && (a.sin_rs % VEC == 0) &&
                  ((reinterpret_cast<uintptr_t>(a.cos) & 31) == 0) &&
                  ((reinterpret_cast<uintptr_t>(a.sin) & 31) == 0);
    if (a.pos3) vec_ok = vec_ok && (a.sec1 % VEC == 0) && (a.sec2 % VEC == 0);
    if (a.n_heads_k > 0)
        vec_ok = vec_ok && aligned16(a.K) && (a.k_bs % VEC == 0) && (a.k_hs % VEC == 0) &&
                 (a.k_ss % VEC == 0);
    dim3 grid((unsigned)a.n_rows), block(256);
    if (vec_ok) hipLaunchKernelGGL((rope_vec_kernel<T, TT, NATIVE>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((rope_scalar_kernel<T, TT, NATIVE>), grid, block, 0, st, a);
    return uamd_launch_status();
}

int dispatch(const RopeArgs& a, int q_dtype, int t_dtype, hipStream_t st) {
    if (a.n_rows <= 0) return a.n_rows == 0 ? UAMD_OK : UAMD_ERR_ARG;
    if (a.head_dim <= 0 || (a.head_dim & 1) || a.seqlen <= 0) return UAMD_ERR_ARG;
    if (a.n_rows > 0x7fffffffLL) return UAMD_ERR_ARG;
#define CASE(QD, TD, T, TT, NAT) if (q_dtype == QD && t_dtype == TD) return launch<T, TT, NAT>(a, st);
    CASE(UAMD_BF16, UAMD_BF16, bf16_t, bf16_t, true)
    CASE(UAMD_F16, UAMD_F16, f16_t, f16_t, true)
    CASE(UAMD_BF16, UAMD_F32, bf16_t, float, false)
    CASE(UAMD_F16, UAMD_F32, f16_t, float, false)
    CASE(UAMD_F32, UAMD_F32, float, float, false)
    CASE(UAMD_BF16, UAMD_F16, bf16_t, f16_t, false)
    CASE(UAMD_F16, UAMD_BF16, f16_t, bf16_t, false)
    CASE(UAMD_F32, UAMD_BF16, float, bf16_t, false)
    CASE(UAMD_F32, UAMD_F16, float, f16_t, false)
#undef CASE
    return UAMD_ERR_DTYPE;
}

}  // namespace

// Dense form: Q is [n_rows, n_heads*head_dim] with row stride q_row_stride; pos = row % seqlen.
// Mirrors Fast_RoPE_Embedding (rope_embedding.py:169-261). backward=1 negates sin.
extern "C" int uamd_rope_embedding(void* Q, int64_t q_row_stride, const void* cos,
                                   int64_t cos_row_stride, const void* sin, int64_t sin_row_stride,
                                   int64_t n_rows, int seqlen, int n_heads, int head_dim,
                                   int backward, int q_dtype, int table_dtype, void* stream) {
    RopeArgs a;
    a.Q = Q; a.q_bs = (int64_t)seqlen * q_row_stride; a.q_hs = head_dim; a.q_ss = q_row_stride;
    a.K = nullptr; a.k_bs = a.k_hs = a.k_ss = 0;
    a.cos = cos; a.cos_rs = cos_row_stride; a.sin = sin; a.sin_rs = sin_row_stride;
    a.idx = nullptr; a.n_rows = n_rows; a.seqlen = seqlen; a.n_heads_q = n_heads;
    a.n_heads_k = 0; a.head_dim = head_dim; a.backward = backward;
    a.pos3 = nullptr; a.sec1 = a.sec2 = 0;
    return dispatch(a, q_dtype, table_dtype, (hipStream_t)stream);
}

// Q/K form: strided [batch, heads, seqlen, head_dim] views (element strides), optional
// per-token int32 gather indices (NULL -> row % seqlen). Mirrors Fast_RoPE_Embedding_QK
// (rope_embedding.py:283-399).
extern "C" int uamd_rope_embedding_qk(void* Q, int64_t q_batch_stride, int64_t q_head_stride,
                                      int64_t q_seq_stride, void* K, int64_t k_batch_stride,
                                      int64_t k_head_stride, int64_t k_seq_stride, const void* cos,
                                      int64_t cos_row_stride, const void* sin,
                                      int64_t sin_row_stride, const int32_t* rope_indices,
                                      int batch, int seqlen, int n_heads_q, int n_heads_k,
                                      int head_dim, int backward, int q_dtype, int table_dtype,
                                      void* stream) {
    RopeArgs a;
    a.Q = Q; a.q_bs = q_batch_stride; a.q_hs = q_head_stride; a.q_ss = q_seq_stride;
    a.K = K; a.k_bs = k_batch_stride; a.k_hs = k_head_stride; a.k_ss = k_seq_stride;
    a.cos = cos; a.cos_rs = cos_row_stride; a.sin = sin; a.sin_rs = sin_row_stride;
    a.idx = rope_indices; a.n_rows = (int64_t)batch * seqlen; a.seqlen = seqlen;
    a.n_heads_q = n_heads_q; a.n_heads_k = K ? n_heads_k : 0; a.head_dim = head_dim;
    a.backward = backward;
    a.pos3 = nullptr; a.sec1 = a.sec2 = 0;
    return dispatch(a, q_dtype, table_dtype, (hipStream_t)stream);
}

// Multimodal RoPE (Qwen2-VL / Qwen2.5-VL "mrope", BASELINE config 4): the reference has no kernel for it (its VLM
// path goes through the unsloth_zoo compiler); semantics = transformers' apply_multimodal_rotary_pos_emb
// (models/qwen2_vl/modeling_qwen2_vl.py): positions3 int32 [3, batch*seqlen] = (temporal, height, width) position of
// every token, mrope_section = (s_t, s_h, s_w) rotary pairs with s_t + s_h + s_w = head_dim / 2: pair j < s_t rotates
// by the temporal position, s_t <= j < s_t + s_h by the height position, the rest by the width position. Same
// in-place strided Q/K contract, same cos/sin table ([>= max position, >= head_dim/2]) as uamd_rope_embedding_qk.
extern "C" int uamd_rope_embedding_qk_mrope(void* Q, int64_t q_batch_stride, int64_t q_head_stride,
                                            int64_t q_seq_stride, void* K, int64_t k_batch_stride,
                                            int64_t k_head_stride, int64_t k_seq_stride, const void* cos,
                                            int64_t cos_row_stride, const void* sin, int64_t sin_row_stride,
                                            const int32_t* positions3, int section_t, int section_h, int batch,
                                            int seqlen, int n_heads_q, int n_heads_k, int head_dim, int backward,
                                            int q_dtype, int table_dtype, void* stream) {
    if (!positions3 || section_t < 0 || section_h < 0 || section_t + section_h > head_dim / 2) return UAMD_ERR_ARG;
    RopeArgs a;
    a.Q = Q; a.q_bs = q_batch_stride; a.q_hs = q_head_stride; a.q_ss = q_seq_stride;
    a.K = K; a.k_bs = k_batch_stride; a.k_hs = k_head_stride; a.k_ss = k_seq_stride;
    a.cos = cos; a.cos_rs = cos_row_stride; a.sin = sin; a.sin_rs = sin_row_stride;
    a.idx = nullptr; a.n_rows = (int64_t)batch * seqlen; a.seqlen = seqlen;
    a.n_heads_q = n_heads_q; a.n_heads_k = K ? n_heads_k : 0; a.head_dim = head_dim;
    a.backward = backward;
    a.pos3 = positions3; a.sec1 = section_t; a.sec2 = section_t + section_h;
    return dispatch(a, q_dtype, table_dtype, (hipStream_t)stream);
}
