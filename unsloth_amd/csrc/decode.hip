// Single-token decode kernels for gfx950: NF4 / 16-bit GEMV with the LoRA term, fused RoPE + KV-cache append, and
// split-KV ("flash decoding") GQA attention over the cache.
//
// Where they sit on the reference's path (decode): LlamaAttention_fast_forward_inference,
// fast_swiglu_inference and LlamaModel_fast_forward_inference (unsloth/models/llama.py:352-569, :572-606, :1249-1364),
// which run every linear of a one-token step through fast_linear_forward -> fast_gemv
// (unsloth/kernels/utils.py:1082-1125, :872-977: bitsandbytes' cgemm_4bit_inference_naive_{fp16,bf16} after a separate
// cdequantize_blockwise_fp32 launch for the nested absmax), RoPE as six in-place torch ops on temporaries
// (llama.py:468-490), the cache append as two permuted copies (:494-497) and attention as matmul / softmax / matmul
// over the whole cache (:533-543).
//
// All of it is HBM- and launch-bound (one token reads every weight once: 0.516 B/param NF4, 2 B/param 16-bit), so:
//   * GEMV: one wave owns whole output rows; a lane's 16-byte load covers 32 NF4 codes (8 dense elements) of the row and
//     always the SAME columns, so its slice of x stays in registers (packed 16-bit pairs) for every row the wave visits;
//     decode + multiply-accumulate is one LDS lookup and one v_dot2c_f32_{bf16,f16} per BYTE of NF4: the lookup table
//     maps a byte to its two decoded values as a packed pair, replicated 32 x in LDS so that lane l only ever touches
//     bank l % 32 (a 256-entry table hit with random indices would serialise 3-4 x on bank conflicts). The nested
//     absmax is decoded in the same kernel (no second launch), applied once per 32 codes in fp32; the LoRA term
//     s * B (A x) and the bias are folded into the wave reduction (lane r adds s * B[n][r] * t[r]). Several
//     projections that share x (q|k|v, gate|up) are ONE launch.
//   * RoPE + append: one launch rotates the new q and k (same arithmetic as the training kernel: fp32, one rounding)
//     and writes k, v at position kv_len[b] of the cache [B, Hk, S_max, D] -- positions come from DEVICE memory, so the
//     whole step is replayable as a hipGraph.
//   * attention: grid (split, kv head, batch); the G query heads of a KV head share every K / V row read; each split
//     keeps an online-softmax partial (m, l, o[D]) that a second tiny kernel combines. The split count is fixed at
//     capture time; splits past the current length exit immediately.
#include "common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

namespace {

__constant__ float kNF4d[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

template <typename T> struct Dot2;
template <> struct Dot2<bf16_t> {
    static __device__ __forceinline__ float run(uint32_t a, uint32_t b, float c) {
        union { uint32_t u; bf16x2_t v; } x, y;
        x.u = a; y.u = b;
        return __builtin_amdgcn_fdot2_f32_bf16(x.v, y.v, c, false);
    }
};
template <> struct Dot2<f16_t> {
    static __device__ __forceinline__ float run(uint32_t a, uint32_t b, float c) {
        union { uint32_t u; f16x2_t v; } x, y;
        x.u = a; y.u = b;
        return __builtin_amdgcn_fdot2(x.v, y.v, c, false);
    }
};

template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    union { T h[2]; uint32_t u; } v;
    v.h[0] = from_f32<T>(lo);
    v.h[1] = from_f32<T>(hi);
    return v.u;
}

#define UAMD_GEMV_MAX_GROUPS 4
struct GemvArgs {
    const void* x;
    int K, n_groups, blocksize, total_rows;
    int row_start[UAMD_GEMV_MAX_GROUPS + 1];
    uamd_gemv_group g[UAMD_GEMV_MAX_GROUPS];
};

// XREGS: registers holding this lane's slice of x = (iterations over K) x (pairs per 16-byte load)
template <typename T, bool NF4, int XREGS>
__global__ void __launch_bounds__(256) gemv_kernel(GemvArgs p) {
    constexpr int PAIRS = NF4 ? 16 : 4;              // 16-bit pairs of x per 16-byte weight load
    constexpr int ELEMS = 2 * PAIRS;                 // columns per lane per iteration
    constexpr int NIT = XREGS / PAIRS;
    __shared__ uint32_t lut2[NF4 ? 256 * 32 : 1];    // [byte][copy]: both decoded values of a byte, packed
    __shared__ float code2[NF4 ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;
    if (NF4) {
        for (int i = tid; i < 256 * 32; i += 256) {
            const int e = i >> 5;
            lut2[i] = pack2<T>(kNF4d[e >> 4], kNF4d[e & 15]);      // high nibble = even element
        }
        code2[tid] = p.g[0].code2 ? p.g[0].code2[tid] : 0.f;       // one nested map per launch (host checks)
    }
    // this lane's columns: k0(i) = (i * 64 + lane) * ELEMS
    uint32_t xr[NIT][PAIRS];
    const T* xp = (const T*)p.x;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int k0 = (i * 64 + lane) * ELEMS;
#pragma unroll
        for (int q = 0; q < PAIRS / 4; ++q) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k0 + q * 8 < K) v = *reinterpret_cast<const uint4*>(xp + k0 + q * 8);    // K % 8 == 0 (host)
            xr[i][4 * q + 0] = v.x; xr[i][4 * q + 1] = v.y; xr[i][4 * q + 2] = v.z; xr[i][4 * q + 3] = v.w;
        }
    }
    if (NF4) __syncthreads();
    const uint32_t* lut_lane = lut2 + (lane & 31);

    const int nwaves = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + wave; row < p.total_rows; row += nwaves) {
        int gi = 0;
#pragma unroll
        for (int i = 1; i < UAMD_GEMV_MAX_GROUPS; ++i)
            if (i < p.n_groups && row >= p.row_start[i]) gi = i;
        const uamd_gemv_group& g = p.g[gi];
        const int n = row - p.row_start[gi];
        // all loads of the row first (they are independent), then the arithmetic
        uint4 w[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int k0 = (i * 64 + lane) * ELEMS;
            w[i] = make_uint4(0, 0, 0, 0);
            if (k0 < K) {
                if (NF4) {
                    const int64_t e0 = (int64_t)n * K + k0;
                    const uamd_u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const uamd_u32x4*>((const uint8_t*)g.W + (e0 >> 1)));
                    w[i] = make_uint4(r[0], r[1], r[2], r[3]);
                } else {
                    w[i] = *reinterpret_cast<const uint4*>((const T*)g.W + (int64_t)n * g.ldw + k0);
                }
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int k0 = (i * 64 + lane) * ELEMS;
            if (k0 < K) {
                const uint32_t ww[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
                float local = 0.f;
                if (NF4) {
                    const int64_t blk = ((int64_t)n * K + k0) / p.blocksize;
                    const float a = g.absmax_f32 ? g.absmax_f32[blk]
                                                 : code2[g.absmax_u8[blk]] * g.absmax2[blk / g.blocksize2] + g.offset;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t byte = (ww[q] >> (8 * b)) & 0xffu;
                            local = Dot2<T>::run(lut_lane[byte * 32], xr[i][4 * q + b], local);
                        }
                    acc += a * local;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) local = Dot2<T>::run(ww[q], xr[i][q], local);
                    acc += local;
                }
            }
        }
        // LoRA: lane r adds s * B[n][r] * t[r]  (t = A x, fp32, from the preceding GEMV launch over the A rows)
        if (g.lora_t && lane < g.R) {
            const float bv = g.lora_b_f32 ? ((const float*)g.lora_b)[(int64_t)n * g.ld_lb + lane]
                                          : to_f32(((const T*)g.lora_b)[(int64_t)n * g.ld_lb + lane]);
            acc += g.lora_scale * bv * g.lora_t[lane];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (g.bias) acc += to_f32(((const T*)g.bias)[n]);
            if (g.y_f32) ((float*)g.y)[n] = acc;
            else ((T*)g.y)[n] = from_f32<T>(acc);
        }
    }
}

template <typename T, bool NF4>
int launch_gemv(const GemvArgs& a, hipStream_t st) {
    constexpr int ELEMS = NF4 ? 32 : 8;
    const int nit = (a.K + 64 * ELEMS - 1) / (64 * ELEMS);
    const int pairs = NF4 ? 16 : 4;
    int blocks = (a.total_rows + 7) / 8;                 // >= 2 rows per wave amortise the table / x setup
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (nit * pairs <= 32) hipLaunchKernelGGL((gemv_kernel<T, NF4, 32>), dim3(blocks), dim3(256), 0, st, a);
    else if (nit * pairs <= 64) hipLaunchKernelGGL((gemv_kernel<T, NF4, 64>), dim3(blocks), dim3(256), 0, st, a);
    else if (nit * pairs <= 128) hipLaunchKernelGGL((gemv_kernel<T, NF4, 128>), dim3(blocks), dim3(256), 0, st, a);
    else return UAMD_ERR_ARG;                            // K > 16384 (NF4) / 4096 * 4 (dense): host splits K
    return uamd_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE (rotate-half, the training kernel's arithmetic: fp32 products, one rounding) on the new token's q and k, and the
// append of k, v to the cache at position kv_len[b]. qkv: [B, (Hq + 2 Hk) D] as the fused q|k|v GEMV wrote it.
template <typename T>
__global__ void __launch_bounds__(256) rope_append_kernel(T* __restrict__ qkv, int64_t ld_qkv, const T* __restrict__ cos_t,
                                                          const T* __restrict__ sin_t, int64_t ld_cs,
                                                          const int* __restrict__ kv_len, const int* __restrict__ rope_pos,
                                                          T* __restrict__ kc, T* __restrict__ vc, int64_t c_sb,
                                                          int64_t c_sh, int Hq, int Hk, int D, int s_max) {
    const int b = blockIdx.x;
    const int half = D >> 1;
    const int len = kv_len[b];
    const int pos = rope_pos ? rope_pos[b] : len;
    T* row = qkv + (int64_t)b * ld_qkv;
    const int nrot = (Hq + Hk) * half;                  // (head, j) pairs to rotate
    for (int i = threadIdx.x; i < nrot; i += 256) {
        const int h = i / half, j = i - h * half;
        T* v = row + (int64_t)h * D;                    // q heads then k heads are contiguous in the fused row
        const float c = to_f32(cos_t[(int64_t)pos * ld_cs + j]), s = to_f32(sin_t[(int64_t)pos * ld_cs + j]);
        const float x1 = to_f32(v[j]), x2 = to_f32(v[j + half]);
        const T r1 = from_f32<T>(x1 * c - x2 * s), r2 = from_f32<T>(x2 * c + x1 * s);
        v[j] = r1;
        v[j + half] = r2;
        if (h >= Hq && len < s_max) {
            T* kd = kc + (int64_t)b * c_sb + (int64_t)(h - Hq) * c_sh + (int64_t)len * D;
            kd[j] = r1;
            kd[j + half] = r2;
        }
    }
    if (len < s_max) {
        const T* vsrc = row + (int64_t)(Hq + Hk) * D;
        for (int i = threadIdx.x; i < Hk * D; i += 256) {
            const int h = i / D, j = i - h * D;
            vc[(int64_t)b * c_sb + (int64_t)h * c_sh + (int64_t)len * D + j] = vsrc[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-KV decode attention, D = 128. Block (split, kvh, b) = 4 waves; a wave-load covers 4 keys (16 lanes x 16 B
// per key); every 16-lane group keeps, for each of the G query heads, an online-softmax partial over its keys with
// the output restricted to the lane's 8 head-dim columns. Partials: [B, Hq, nsplit, D + 2] fp32 (o[D], m, l).
constexpr int DD = 128;
template <typename T, int G>
__global__ void __launch_bounds__(256) attn_decode_kernel(const T* __restrict__ q, int64_t q_sb, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, int64_t c_sb, int64_t c_sh,
                                                          const int* __restrict__ kv_len, float* __restrict__ part,
                                                          int Hq, int nsplit, int split_keys, int window,
                                                          float scale_log2, int len_add) {
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 4, l16 = lane & 15;           // key slot of the wave-load, 8-column slice
    const int len = kv_len[b] + len_add;                  // keys 0 .. len-1 are valid (the new token included)
    const int first = (window > 0 && len > window) ? len - window : 0;
    const int s0 = split * split_keys, s1 = min(s0 + split_keys, len);
    __shared__ float red[4][G][2];                        // [wave][head][m, l]
    __shared__ float red_o[4][G][DD];
    float m[G], l[G], o[G][8];
    float qv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY; l[g] = 0.f;
        const T* qp = q + (int64_t)b * q_sb + (int64_t)(kvh * G + g) * DD + l16 * 8;
        const Vec16<T> v = ld16(qp);
#pragma unroll
        for (int j = 0; j < 8; ++j) { qv[g][j] = to_f32(v.e[j]) * scale_log2; o[g][j] = 0.f; }
    }
    const T* kb = kc + (int64_t)b * c_sb + (int64_t)kvh * c_sh;
    const T* vb = vc + (int64_t)b * c_sb + (int64_t)kvh * c_sh;
    for (int k0 = max(s0, first & ~15) + wave * 4; k0 < s1; k0 += 16) {
        const int key = k0 + grp;
        const bool valid = key < s1 && key >= first;
        const int kl = valid ? key : (len > 0 ? len - 1 : 0);
        const Vec16<T> kk = ld16(kb + (int64_t)kl * DD + l16 * 8);
        const Vec16<T> vv = ld16(vb + (int64_t)kl * DD + l16 * 8);
        float s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a += qv[g][j] * to_f32(kk.e[j]);
            a += __shfl_xor(a, 8, 64);
            a += __shfl_xor(a, 4, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 1, 64);
            s[g] = valid ? a : -INFINITY;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mn = fmaxf(m[g], s[g]);
            const float mr = mn == -INFINITY ? 0.f : mn;
            const float alpha = __builtin_amdgcn_exp2f(m[g] - mr);
            const float pe = __builtin_amdgcn_exp2f(s[g] - mr);
            m[g] = mn;
            l[g] = l[g] * alpha + pe;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * alpha + pe * to_f32(vv.e[j]);
        }
    }
    // combine: the 4 key slots of a wave by shuffles (lanes with the same column slice), then the 4 waves through LDS
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float mo = __shfl_xor(m[g], off, 64), lo = __shfl_xor(l[g], off, 64);
            const float mn = fmaxf(m[g], mo);
            const float mr = mn == -INFINITY ? 0.f : mn;
            const float a = __builtin_amdgcn_exp2f(m[g] - mr), c = __builtin_amdgcn_exp2f(mo - mr);
            l[g] = l[g] * a + lo * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * a + __shfl_xor(o[g][j], off, 64) * c;
            m[g] = mn;
        }
        if (lane < 16) {
            if (l16 == 0) { red[wave][g][0] = m[g]; red[wave][g][1] = l[g]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) red_o[wave][g][l16 * 8 + j] = o[g][j];
        }
    }
    __syncthreads();
    for (int i = tid; i < G * DD; i += 256) {
        const int g = i / DD, d = i - g * DD;
        float mm = -INFINITY;
#pragma unroll
        for (int s = 0; s < 4; ++s) mm = fmaxf(mm, red[s][g][0]);
        const float mr = mm == -INFINITY ? 0.f : mm;
        float ll = 0.f, oo = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a = __builtin_amdgcn_exp2f(red[s][g][0] - mr);
            ll += red[s][g][1] * a;
            oo += red_o[s][g][d] * a;
        }
        float* pp = part + (((int64_t)b * Hq + kvh * G + g) * nsplit + split) * (DD + 2);
        pp[d] = oo;
        if (d == 0) { pp[DD] = mm; pp[DD + 1] = ll; }
    }
}

template <typename T>
__global__ void __launch_bounds__(128) attn_decode_combine_kernel(const float* __restrict__ part, T* __restrict__ out,
                                                                  int64_t o_sb, int Hq, int nsplit) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* pp = part + ((int64_t)b * Hq + h) * nsplit * (DD + 2);
    float mm = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, pp[s * (DD + 2) + DD]);
    const float mr = mm == -INFINITY ? 0.f : mm;
    float ll = 0.f, oo = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float a = __builtin_amdgcn_exp2f(pp[s * (DD + 2) + DD] - mr);
        ll += pp[s * (DD + 2) + DD + 1] * a;
        oo += pp[s * (DD + 2) + d] * a;
    }
    out[(int64_t)b * o_sb + (int64_t)h * DD + d] = from_f32<T>(ll > 0.f ? oo / ll : 0.f);
}

}  // namespace

// y_g[n] = W_g[n, :] . x  (+ s_g * B_g[n, :] . t_g + bias_g[n]) for up to 4 row groups sharing x (one token).
// nf4 != 0: W_g is bitsandbytes-format NF4 (packed [N, K/2], absmax per `blocksize` codes, nested when absmax_u8 is
// given: all groups of a launch share code2). Replaces fast_gemv / the bsz == 1 branch of fast_linear_forward
// (unsloth/kernels/utils.py:872-977, :1082-1125) in ONE launch instead of cdequantize_blockwise_fp32 +
// cgemm_4bit_inference_naive + mv + addmv.
extern "C" int uamd_gemv(const void* x, int K, const uamd_gemv_group* groups, int n_groups, int nf4, int blocksize,
                         int dtype, void* stream) {
    if (!x || !groups || n_groups < 1 || n_groups > UAMD_GEMV_MAX_GROUPS || K <= 0) return UAMD_ERR_ARG;
    if ((K & 7) || !aligned16(x)) return UAMD_ERR_ALIGN;
    if (nf4 && (blocksize < 32 || (blocksize & 31) || (K & 31))) return UAMD_ERR_ARG;
    GemvArgs a;
    a.x = x; a.K = K; a.n_groups = n_groups; a.blocksize = blocksize;
    int rows = 0;
    for (int i = 0; i < UAMD_GEMV_MAX_GROUPS; ++i) {
        a.row_start[i] = rows;
        if (i < n_groups) {
            const uamd_gemv_group& g = groups[i];
            if (!g.W || !g.y || g.N <= 0) return UAMD_ERR_ARG;
            if (!aligned16(g.W)) return UAMD_ERR_ALIGN;
            if (nf4) {
                if (!g.absmax_f32 && !(g.absmax_u8 && g.code2 && g.absmax2 && g.blocksize2 > 0)) return UAMD_ERR_ARG;
                if (g.absmax_u8 && g.code2 != groups[0].code2) return UAMD_ERR_ARG;
            } else if (g.ldw & 7) {
                return UAMD_ERR_ALIGN;
            }
            if (g.lora_t && (!g.lora_b || g.R <= 0 || g.R > 64)) return UAMD_ERR_ARG;
            a.g[i] = g;
            rows += g.N;
        } else {
            a.g[i] = groups[0];
        }
    }
    a.row_start[UAMD_GEMV_MAX_GROUPS] = rows;
    a.total_rows = rows;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) return nf4 ? launch_gemv<bf16_t, true>(a, st) : launch_gemv<bf16_t, false>(a, st);
    if (dtype == UAMD_F16) return nf4 ? launch_gemv<f16_t, true>(a, st) : launch_gemv<f16_t, false>(a, st);
    return UAMD_ERR_DTYPE;
}

// RoPE on the new token's q, k (in place in the fused qkv row) + append of k, v at cache position kv_len[b]
// (replaces llama.py:468-497). cos / sin: [positions, >= D/2] tables; rope_pos NULL = kv_len.
extern "C" int uamd_rope_kv_append(void* qkv, int64_t ld_qkv, const void* cos_t, const void* sin_t, int64_t ld_cs,
                                   const int* kv_len, const int* rope_pos, void* k_cache, void* v_cache,
                                   int64_t cache_sb, int64_t cache_sh, int B, int Hq, int Hk, int D, int s_max,
                                   int dtype, void* stream) {
    if (!qkv || !cos_t || !sin_t || !kv_len || !k_cache || !v_cache || B <= 0 || Hq <= 0 || Hk <= 0 || D <= 0 || (D & 1))
        return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16)
        hipLaunchKernelGGL((rope_append_kernel<bf16_t>), dim3(B), dim3(256), 0, st, (bf16_t*)qkv, ld_qkv, (const bf16_t*)cos_t,
                           (const bf16_t*)sin_t, ld_cs, kv_len, rope_pos, (bf16_t*)k_cache, (bf16_t*)v_cache, cache_sb,
                           cache_sh, Hq, Hk, D, s_max);
    else if (dtype == UAMD_F16)
        hipLaunchKernelGGL((rope_append_kernel<f16_t>), dim3(B), dim3(256), 0, st, (f16_t*)qkv, ld_qkv, (const f16_t*)cos_t,
                           (const f16_t*)sin_t, ld_cs, kv_len, rope_pos, (f16_t*)k_cache, (f16_t*)v_cache, cache_sb,
                           cache_sh, Hq, Hk, D, s_max);
    else
        return UAMD_ERR_DTYPE;
    return uamd_launch_status();
}

// out[b, h, :] = softmax(q[b, h] . K[b, h / G, first..len) * scale) V  over the cache (len = kv_len[b] + len_add;
// window > 0: only the last `window` keys). partials: fp32 workspace [B, Hq, nsplit, D + 2]. Replaces
// llama.py:499-543 (expand + matmul + softmax + matmul, or SDPA).
extern "C" int uamd_attn_decode(const void* q, int64_t q_sb, const void* k_cache, const void* v_cache, int64_t cache_sb,
                                int64_t cache_sh, const int* kv_len, int len_add, float* partials, void* out,
                                int64_t out_sb, int B, int Hq, int Hk, int D, int nsplit, int split_keys, int window,
                                float scale, int dtype, void* stream) {
    if (!q || !k_cache || !v_cache || !kv_len || !partials || !out || B <= 0 || Hq <= 0 || Hk <= 0 || Hq % Hk) return UAMD_ERR_ARG;
    if (D != DD || nsplit <= 0 || split_keys <= 0 || (split_keys & 15)) return UAMD_ERR_ARG;
    if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || (q_sb & 7) || (cache_sb & 7) || (cache_sh & 7)) return UAMD_ERR_ALIGN;
    const int G = Hq / Hk;
    const float sl2 = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nsplit, (unsigned)Hk, (unsigned)B);
#define UAMD_DECODE_LAUNCH(TT, GG)                                                                                       \
    hipLaunchKernelGGL((attn_decode_kernel<TT, GG>), grid, dim3(256), 0, st, (const TT*)q, q_sb, (const TT*)k_cache,      \
                       (const TT*)v_cache, cache_sb, cache_sh, kv_len, partials, Hq, nsplit, split_keys, window, sl2, len_add)
    if (dtype == UAMD_BF16) {
        if (G == 1) UAMD_DECODE_LAUNCH(bf16_t, 1); else if (G == 2) UAMD_DECODE_LAUNCH(bf16_t, 2);
        else if (G == 4) UAMD_DECODE_LAUNCH(bf16_t, 4); else if (G == 8) UAMD_DECODE_LAUNCH(bf16_t, 8); else return UAMD_ERR_ARG;
    } else if (dtype == UAMD_F16) {
        if (G == 1) UAMD_DECODE_LAUNCH(f16_t, 1); else if (G == 2) UAMD_DECODE_LAUNCH(f16_t, 2);
        else if (G == 4) UAMD_DECODE_LAUNCH(f16_t, 4); else if (G == 8) UAMD_DECODE_LAUNCH(f16_t, 8); else return UAMD_ERR_ARG;
    } else {
        return UAMD_ERR_DTYPE;
    }
#undef UAMD_DECODE_LAUNCH
    int rc = uamd_launch_status();
    if (rc) return rc;
    if (dtype == UAMD_BF16)
        hipLaunchKernelGGL((attn_decode_combine_kernel<bf16_t>), dim3(Hq, B), dim3(DD), 0, st, partials, (bf16_t*)out, out_sb, Hq, nsplit);
    else
        hipLaunchKernelGGL((attn_decode_combine_kernel<f16_t>), dim3(Hq, B), dim3(DD), 0, st, partials, (f16_t*)out, out_sb, Hq, nsplit);
    return uamd_launch_status();
}
