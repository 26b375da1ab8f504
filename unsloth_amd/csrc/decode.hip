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

// Reductions on the VALU only (DPP), no LDS round trips: a ds_bpermute-based __shfl_xor costs ~100+ cycles of latency per
// step, and a decode step is nothing but latency. row16_sum: every lane ends with the sum over its 16-lane row
// (quad swaps, then the two mirror patterns pair each lane with the partial sum it is missing).
#define UAMD_DPP_ADD(v, CTRL) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v)), (CTRL), 0xf, 0xf, true)))
__device__ __forceinline__ float row16_sum(float v) {
    v = UAMD_DPP_ADD(v, 0xb1);      // quad_perm [1,0,3,2]
    v = UAMD_DPP_ADD(v, 0x4e);      // quad_perm [2,3,0,1]
    v = UAMD_DPP_ADD(v, 0x141);     // row_half_mirror
    v = UAMD_DPP_ADD(v, 0x140);     // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {             // wave-uniform total of all 64 lanes
    v = row16_sum(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

#define UAMD_GEMV_MAX_GROUPS 4
struct GemvArgs {
    const void* x;
    int K, n_groups, bs_shift, total_rows;         // blocksize = 1 << bs_shift
    int row_start[UAMD_GEMV_MAX_GROUPS + 1];
    uamd_gemv_group g[UAMD_GEMV_MAX_GROUPS];
    uamd_gemv_prologue pro;                        // mode 0 / a_rows NULL = the plain kernel
};

// One block = 8 waves sharing the decode table and the token x in LDS; a wave takes RB ADJACENT rows per trip and
// issues every global load of the trip (weights, absmax codes) before anything else -- on the first trip even before
// the table is built -- so 4-8 KB per wave are in flight while the fixed costs are paid. NIT = iterations over K.
constexpr int GEMV_THREADS = 512;
template <typename T, bool NF4, int NIT, int RB>
__global__ void __launch_bounds__(GEMV_THREADS) gemv_kernel(GemvArgs p) {
    constexpr int PAIRS = NF4 ? 16 : 4;              // 16-bit pairs of x per 16-byte weight load
    constexpr int ELEMS = 2 * PAIRS;                 // columns per lane per iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char gemv_smem[];
    // layout: x [NIT * 64 * ELEMS] T | code2 [4][256] float | lut2 [256][32] u32 (NF4 only)
    T* xs = reinterpret_cast<T*>(gemv_smem);
    float* code2 = reinterpret_cast<float*>(gemv_smem + NIT * 64 * ELEMS * sizeof(T));
    uint32_t* lut2 = reinterpret_cast<uint32_t*>(code2 + UAMD_GEMV_MAX_GROUPS * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K;
    // columns past K: x is zero there, so the lane may read any valid address instead (no branches in the row loop)
    int koff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int k0 = (i * 64 + lane) * ELEMS;
        koff[i] = k0 < K ? k0 : 0;
    }
    const int nwaves = gridDim.x * (GEMV_THREADS / 64);
    int gis[RB], ns[RB];
    float direct[RB];                                // 1: single-level fp32 absmax, 0: nested
    uint4 w[RB][NIT];
    uint32_t a8[RB][NIT];
    float a2[RB][NIT];
    auto load_rows = [&](int row0) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = min(row0 + r, p.total_rows - 1);         // a duplicate of the last row, not stored
            int gi = 0;
#pragma unroll
            for (int i = 1; i < UAMD_GEMV_MAX_GROUPS; ++i)
                if (i < p.n_groups && row >= p.row_start[i]) gi = i;
            gis[r] = gi;
            ns[r] = row - p.row_start[gi];
            const uamd_gemv_group& g = p.g[gi];
            direct[r] = g.absmax_f32 ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                a8[r][i] = 0;
                a2[r][i] = 0.f;
                if (NF4) {
                    const int64_t e0 = (int64_t)ns[r] * K + koff[i];
                    const uamd_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const uamd_u32x4*>((const uint8_t*)g.W + (e0 >> 1)));
                    w[r][i] = make_uint4(v[0], v[1], v[2], v[3]);
                    const int64_t blk = e0 >> p.bs_shift;
                    if (g.absmax_f32) {
                        a2[r][i] = g.absmax_f32[blk];
                    } else {
                        a8[r][i] = g.absmax_u8[blk];
                        a2[r][i] = g.absmax2[blk >> g.blocksize2];       // (the host stores log2(blocksize2) here)
                    }
                } else {
                    w[r][i] = *reinterpret_cast<const uint4*>((const T*)g.W + (int64_t)ns[r] * g.ldw + koff[i]);
                }
            }
        }
    };
    int row0 = (blockIdx.x * (GEMV_THREADS / 64) + wave_u) * RB;
    if (row0 < p.total_rows) load_rows(row0);
    // ---- block setup while those loads fly: x (zero-padded; optionally PRODUCED here: SwiGLU of two vectors, or
    //      residual add + RMSNorm), t = A x of the LoRA factors, the nested-absmax maps, the byte -> value-pair table
    float* tl = reinterpret_cast<float*>(gemv_smem + NIT * 64 * ELEMS * sizeof(T) + UAMD_GEMV_MAX_GROUPS * 256 * 4 +
                                          (NF4 ? 256 * 32 * 4 : 0));          // [64 ranks x 4 groups] + 8 reduction slots
    {
        const T* xp = (const T*)p.x;
        const int nvec = NIT * 64 * ELEMS / 8;
        const int mode = p.pro.mode;
        if (mode == 0) {
            for (int v = tid; v < nvec; v += GEMV_THREADS) {
                uint4 val = make_uint4(0, 0, 0, 0);
                if (v * 8 < K) val = *reinterpret_cast<const uint4*>(xp + v * 8);        // K % 8 == 0 (host)
                reinterpret_cast<uint4*>(xs)[v] = val;
            }
        } else if (mode == 1) {
            // x = (e * sigmoid(e)).to(T) * g: the SwiGLU of fast_swiglu_inference (llama.py:572-606), rounding points of
            // the training kernel (csrc/glu.hip); p.x = e (gate), pro.x2 = g (up)
            const T* gp = (const T*)p.pro.x2;
            for (int v = tid; v < nvec; v += GEMV_THREADS) {
                union { uint4 r; T e[8]; } a, b, o;
                o.r = make_uint4(0, 0, 0, 0);
                if (v * 8 < K) {
                    a.r = *reinterpret_cast<const uint4*>(xp + v * 8);
                    b.r = *reinterpret_cast<const uint4*>(gp + v * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float e = to_f32(a.e[j]);
                        const float f = e * (1.0f / (1.0f + __expf(-e)));
                        o.e[j] = from_f32<T>(round_to<T>(f) * to_f32(b.e[j]));
                    }
                }
                reinterpret_cast<uint4*>(xs)[v] = o.r;
            }
        } else {
            // x = rmsnorm(h) * w, h = T(a + res) (a = p.x may be NULL: h = res): fast_rms_layernorm_inference after the
            // residual add of the decoder layer (llama.py:352-606), rounding points of csrc/rms_layernorm.hip. Every block
            // normalises the whole row for itself; block 0 also writes h (the next residual) to pro.h_out.
            const T* rp = (const T*)p.pro.res;
            T* hp = (T*)p.pro.h_out;
            float ss = 0.f;
            for (int v = tid; v < nvec; v += GEMV_THREADS) {
                union { uint4 r; T e[8]; } a, b;
                b.r = make_uint4(0, 0, 0, 0);
                if (v * 8 < K) {
                    b.r = *reinterpret_cast<const uint4*>(rp + v * 8);
                    if (xp) {
                        a.r = *reinterpret_cast<const uint4*>(xp + v * 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) b.e[j] = from_f32<T>(to_f32(a.e[j]) + to_f32(b.e[j]));
                    }
                    if (hp && blockIdx.x == 0) *reinterpret_cast<uint4*>(hp + v * 8) = b.r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = to_f32(b.e[j]); ss += f * f; }
                }
                reinterpret_cast<uint4*>(xs)[v] = b.r;                              // h for now
            }
            ss = wave_sum_dpp(ss);
            if (lane == 0) tl[256 + wave_u] = ss;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_THREADS / 64; ++w) tot += tl[256 + w];
            const float inv = rsqrtf(tot / (float)K + p.pro.eps);
            for (int v = tid; v < nvec; v += GEMV_THREADS) {
                if (v * 8 >= K) continue;
                union { uint4 r; T e[8]; } h;
                h.r = reinterpret_cast<uint4*>(xs)[v];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float normed = to_f32(h.e[j]) * inv;
                    if (p.pro.w_f32) {
                        h.e[j] = from_f32<T>(normed * ((const float*)p.pro.norm_w)[v * 8 + j]);
                    } else {                                        // (x * r).to(W.dtype) * W, product in W's dtype
                        h.e[j] = from_f32<T>(round_to<T>(round_to<T>(normed) * to_f32(((const T*)p.pro.norm_w)[v * 8 + j])));
                    }
                }
                reinterpret_cast<uint4*>(xs)[v] = h.r;
            }
        }
        if (NF4) {
            for (int i = tid; i < UAMD_GEMV_MAX_GROUPS * 256; i += GEMV_THREADS) {
                const int gi = i >> 8;
                code2[i] = (gi < p.n_groups && p.g[gi].code2) ? p.g[gi].code2[i & 255] : 0.f;
            }
            if (tid < 256) {       // thread e builds entry e (high nibble = even element): 32 copies = 8 x 16 bytes
                const uint32_t v = pack2<T>(kNF4d[tid >> 4], kNF4d[tid & 15]);
                uint4* dst = reinterpret_cast<uint4*>(lut2 + tid * 32);
#pragma unroll
                for (int c = 0; c < 8; ++c) dst[c] = make_uint4(v, v, v, v);
            }
        }
    }
    __syncthreads();
    // ---- t = A x for the stacked LoRA A rows of the launch's projections ([Rt, K], activation dtype): every block
    //      computes all of it (Rt <= 256 rows x K: L2-resident, 1-2 us) instead of a launch of its own in front of this one
    if (p.pro.a_rows) {
        const T* Ar = (const T*)p.pro.a_rows;
        const int Rt = p.pro.Rt;
        for (int r = wave_u; r < Rt; r += GEMV_THREADS / 64) {
            float acc = 0.f;
            for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
                union { uint4 r; uint32_t w[4]; } a, xv;
                a.r = *reinterpret_cast<const uint4*>(Ar + (int64_t)r * p.pro.ld_a + k0);
                xv.r = *reinterpret_cast<const uint4*>(xs + k0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = Dot2<T>::run(a.w[j], xv.w[j], acc);
            }
            acc = wave_sum_dpp(acc);
            if (lane == 0) tl[r] = acc;
        }
        __syncthreads();
    }
    const uint32_t* lut_lane = lut2 + (lane & 31);
    const uint4* x_lane = reinterpret_cast<const uint4*>(xs) + lane * (PAIRS / 4);

    for (; row0 < p.total_rows; row0 += nwaves * RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const uamd_gemv_group& g = p.g[gis[r]];
            acc[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const uint32_t ww[4] = {w[r][i].x, w[r][i].y, w[r][i].z, w[r][i].w};
                uint32_t xv[PAIRS];
#pragma unroll
                for (int q = 0; q < PAIRS / 4; ++q) {
                    const uint4 t4 = x_lane[i * 64 * (PAIRS / 4) + q];
                    xv[4 * q] = t4.x; xv[4 * q + 1] = t4.y; xv[4 * q + 2] = t4.z; xv[4 * q + 3] = t4.w;
                }
                if (NF4) {
                    // branch-free (a branch here splits the row into basic blocks and the dot products sink past all of them)
                    const float a_nested = code2[gis[r] * 256 + a8[r][i]] * a2[r][i] + g.offset;
                    const float a = direct[r] * a2[r][i] + (1.f - direct[r]) * a_nested;       // direct is exactly 0 or 1
                    uint32_t dec[16];                        // all 16 table reads of the load first: one LDS latency, not 16
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int b = 0; b < 4; ++b) dec[4 * q + b] = lut_lane[((ww[q] >> (8 * b)) & 0xffu) * 32];
                    float l0 = 0.f, l1 = 0.f;
#pragma unroll
                    for (int j = 0; j < 16; j += 2) {
                        l0 = Dot2<T>::run(dec[j], xv[j], l0);
                        l1 = Dot2<T>::run(dec[j + 1], xv[j + 1], l1);
                    }
                    acc[r] += a * (l0 + l1);
                } else {
                    float l0 = Dot2<T>::run(ww[0], xv[0], 0.f), l1 = Dot2<T>::run(ww[1], xv[1], 0.f);
                    l0 = Dot2<T>::run(ww[2], xv[2], l0);
                    l1 = Dot2<T>::run(ww[3], xv[3], l1);
                    acc[r] += l0 + l1;
                }
                if (NIT > 2) __builtin_amdgcn_sched_barrier(0);    // keep the table / x reads of later iterations from piling up in registers
            }
            // LoRA: lane j adds s * B[n][j] * t[j]  (t = A x, fp32, from the preceding GEMV launch over the A rows)
            if (g.lora_b && g.R > 0 && lane < g.R) {
                const float bv = g.lora_b_f32 ? ((const float*)g.lora_b)[(int64_t)ns[r] * g.ld_lb + lane]
                                              : to_f32(((const T*)g.lora_b)[(int64_t)ns[r] * g.ld_lb + lane]);
                // t from the preceding launch (g.lora_t) or computed by this block (pro.a_rows: rows t_off[g] ..)
                const float tv = p.pro.a_rows ? tl[p.pro.t_off[gis[r]] + lane] : g.lora_t[lane];
                acc[r] += g.lora_scale * bv * tv;
            }
        }
        float tot[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) tot[r] = wave_sum_dpp(acc[r]);
        const int cur = row0;
        int gi_s[RB], n_s[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) { gi_s[r] = gis[r]; n_s[r] = ns[r]; }
        if (row0 + nwaves * RB < p.total_rows) load_rows(row0 + nwaves * RB);        // next trip's loads before the stores
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (cur + r >= p.total_rows) break;
                const uamd_gemv_group& g = p.g[gi_s[r]];
                float v = tot[r];
                if (g.bias) v += to_f32(((const T*)g.bias)[n_s[r]]);
                if (g.y_f32) ((float*)g.y)[n_s[r]] = v;
                else ((T*)g.y)[n_s[r]] = from_f32<T>(v);
            }
        }
    }
}

template <typename T, bool NF4, int NIT, int RB>
int launch_gemv_n(const GemvArgs& a, hipStream_t st) {
    constexpr int ELEMS = NF4 ? 32 : 8;
    const int lds = NIT * 64 * ELEMS * (int)sizeof(T) + UAMD_GEMV_MAX_GROUPS * 256 * 4 + (NF4 ? 256 * 32 * 4 : 0) + (256 + 8) * 4;
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > 48 * 1024 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, NF4, NIT, RB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    const int per_block = (GEMV_THREADS / 64) * RB;
    int blocks = (a.total_rows + per_block - 1) / per_block;     // one trip per wave until the chip is full
    if (blocks > 512) blocks = 512;                               // 2 blocks (16 waves) per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((gemv_kernel<T, NF4, NIT, RB>), dim3(blocks), dim3(GEMV_THREADS), lds, st, a);
    return uamd_launch_status();
}

template <typename T, bool NF4>
int launch_gemv(const GemvArgs& a, hipStream_t st) {
    constexpr int ELEMS = NF4 ? 32 : 8;
    const int nit = (a.K + 64 * ELEMS - 1) / (64 * ELEMS);
    if constexpr (NF4) {                             // K <= 4096 | 8192 | 16384
        if (nit <= 2) return launch_gemv_n<T, true, 2, 4>(a, st);
        if (nit <= 4) return launch_gemv_n<T, true, 4, 2>(a, st);
        if (nit <= 8) return launch_gemv_n<T, true, 8, 1>(a, st);
    } else {                                         // K <= 4096 | 8192 | 16384
        if (nit <= 8) return launch_gemv_n<T, false, 8, 2>(a, st);
        if (nit <= 16) return launch_gemv_n<T, false, 16, 1>(a, st);
        if (nit <= 32) return launch_gemv_n<T, false, 32, 1>(a, st);
    }
    return UAMD_ERR_ARG;                             // larger K: the caller splits it
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE (rotate-half, the training kernel's arithmetic and rounding points) on the new token's q and k, and the
// append of k, v to the cache at position kv_len[b]. qkv: [B, (Hq + 2 Hk) D] as the fused q|k|v GEMV wrote it.
template <typename T>
__global__ void __launch_bounds__(64) rope_append_kernel(T* __restrict__ qkv, int64_t ld_qkv, const T* __restrict__ cos_t,
                                                         const T* __restrict__ sin_t, int64_t ld_cs,
                                                         const int* __restrict__ kv_len, const int* __restrict__ rope_pos,
                                                         T* __restrict__ kc, T* __restrict__ vc, int64_t c_sb,
                                                         int64_t c_sh, int Hq, int Hk, int D, int s_max) {
    const int h = blockIdx.x, b = blockIdx.y;            // h: q heads, then k heads, then v heads of the fused row
    const int half = D >> 1;
    const int len = kv_len[b];
    const int pos = rope_pos ? rope_pos[b] : len;
    T* v = qkv + (int64_t)b * ld_qkv + (int64_t)h * D;
    if (h < Hq + Hk) {
        T* kd = (h >= Hq && len < s_max) ? kc + (int64_t)b * c_sb + (int64_t)(h - Hq) * c_sh + (int64_t)len * D : nullptr;
        for (int j = threadIdx.x; j < half; j += 64) {
            const float c = to_f32(cos_t[(int64_t)pos * ld_cs + j]), s = to_f32(sin_t[(int64_t)pos * ld_cs + j]);
            const float x1 = to_f32(v[j]), x2 = to_f32(v[j + half]);
            // rounding points of the training kernel for 16-bit tables (csrc/rope_embedding.hip rotate<T, NATIVE>, i.e. the
            // reference's Triton arithmetic in the cos / sin dtype): every product and the sum are rounded to T
            const T r1 = from_f32<T>(round_to<T>(x1 * c) - round_to<T>(x2 * s));
            const T r2 = from_f32<T>(round_to<T>(x2 * c) + round_to<T>(x1 * s));
            v[j] = r1;
            v[j + half] = r2;
            if (kd) { kd[j] = r1; kd[j + half] = r2; }
        }
    } else if (len < s_max) {
        T* vd = vc + (int64_t)b * c_sb + (int64_t)(h - Hq - Hk) * c_sh + (int64_t)len * D;
        for (int j = threadIdx.x; j < D; j += 64) vd[j] = v[j];
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
            a = row16_sum(a);                             // the key's 16 lanes all get q . k
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
    // running (max, sum, value) over the splits in a fixed order; 4 splits' loads in flight at a time
    float mm = -INFINITY, ll = 0.f, oo = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 4) {
        float ms[4], ls[4], os[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = min(s0 + j, nsplit - 1);
            ms[j] = s0 + j < nsplit ? pp[s * (DD + 2) + DD] : -INFINITY;
            ls[j] = pp[s * (DD + 2) + DD + 1];
            os[j] = pp[s * (DD + 2) + d];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mn = fmaxf(mm, ms[j]);
            const float mr = mn == -INFINITY ? 0.f : mn;
            const float a = __builtin_amdgcn_exp2f(mm - mr), c = __builtin_amdgcn_exp2f(ms[j] - mr);
            ll = ll * a + ls[j] * c;
            oo = oo * a + os[j] * c;
            mm = mn;
        }
    }
    out[(int64_t)b * o_sb + (int64_t)h * DD + d] = from_f32<T>(ll > 0.f ? oo / ll : 0.f);
}

}  // namespace

// y_g[n] = W_g[n, :] . x  (+ s_g * B_g[n, :] . t_g + bias_g[n]) for up to 4 row groups sharing x (one token).
// nf4 != 0: W_g is bitsandbytes-format NF4 (packed [N, K/2], absmax per `blocksize` codes, nested when absmax_u8 is
// given: all groups of a launch share code2). Replaces fast_gemv / the bsz == 1 branch of fast_linear_forward
// (unsloth/kernels/utils.py:872-977, :1082-1125) in ONE launch instead of cdequantize_blockwise_fp32 +
// cgemm_4bit_inference_naive + mv + addmv.
static int gemv_entry(const void* x, int K, const uamd_gemv_group* groups, int n_groups, int nf4, int blocksize,
                      int dtype, void* stream, const uamd_gemv_prologue* pro) {
    const int mode = pro ? pro->mode : 0;
    if (!groups || n_groups < 1 || n_groups > UAMD_GEMV_MAX_GROUPS || K <= 0 || mode < 0 || mode > 2) return UAMD_ERR_ARG;
    if (!x && mode != 2) return UAMD_ERR_ARG;
    if ((K & 7) || (x && !aligned16(x))) return UAMD_ERR_ALIGN;
    if (mode == 1 && (!pro->x2 || !aligned16(pro->x2))) return UAMD_ERR_ARG;
    if (mode == 2 && (!pro->res || !pro->norm_w || !aligned16(pro->res) || (pro->h_out && !aligned16(pro->h_out)))) return UAMD_ERR_ARG;
    if (pro && pro->a_rows && (pro->Rt <= 0 || pro->Rt > 256 || (pro->ld_a & 7) || !aligned16(pro->a_rows))) return UAMD_ERR_ARG;
    if (nf4 && (blocksize < 32 || (blocksize & 31) || (K & 31))) return UAMD_ERR_ARG;
    GemvArgs a;
    auto log2_exact = [](int v) { int sft = 0; while ((1 << sft) < v) ++sft; return (1 << sft) == v ? sft : -1; };
    a.x = x; a.K = K; a.n_groups = n_groups; a.bs_shift = nf4 ? log2_exact(blocksize) : 0;
    if (a.bs_shift < 0) return UAMD_ERR_ARG;
    int rows = 0;
    for (int i = 0; i < UAMD_GEMV_MAX_GROUPS; ++i) {
        a.row_start[i] = rows;
        if (i < n_groups) {
            const uamd_gemv_group& g = groups[i];
            if (!g.W || !g.y || g.N <= 0) return UAMD_ERR_ARG;
            if (!aligned16(g.W)) return UAMD_ERR_ALIGN;
            if (nf4) {
                if (!g.absmax_f32 && !(g.absmax_u8 && g.code2 && g.absmax2 && g.blocksize2 > 0)) return UAMD_ERR_ARG;
            } else if (g.ldw & 7) {
                return UAMD_ERR_ALIGN;
            }
            if ((g.lora_t || (pro && pro->a_rows && g.lora_b)) && (!g.lora_b || g.R <= 0 || g.R > 64)) return UAMD_ERR_ARG;
            if (pro && pro->a_rows && g.lora_b && (pro->t_off[i] < 0 || pro->t_off[i] + g.R > pro->Rt)) return UAMD_ERR_ARG;
            a.g[i] = g;
            if (!(g.lora_t || (pro && pro->a_rows))) a.g[i].lora_b = nullptr;      // (a B without any t: no LoRA term)
            if (nf4 && !g.absmax_f32) {
                a.g[i].blocksize2 = log2_exact(g.blocksize2);
                if (a.g[i].blocksize2 < 0) return UAMD_ERR_ARG;
            }
            rows += g.N;
        } else {
            a.g[i] = groups[0];
        }
    }
    a.row_start[UAMD_GEMV_MAX_GROUPS] = rows;
    a.total_rows = rows;
    if (pro) a.pro = *pro;
    else {
        a.pro.mode = 0; a.pro.x2 = nullptr; a.pro.res = nullptr; a.pro.norm_w = nullptr; a.pro.h_out = nullptr;
        a.pro.a_rows = nullptr; a.pro.ld_a = 0; a.pro.Rt = 0; a.pro.w_f32 = 0; a.pro.eps = 0.f;
        for (int i = 0; i < UAMD_GEMV_MAX_GROUPS; ++i) a.pro.t_off[i] = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) return nf4 ? launch_gemv<bf16_t, true>(a, st) : launch_gemv<bf16_t, false>(a, st);
    if (dtype == UAMD_F16) return nf4 ? launch_gemv<f16_t, true>(a, st) : launch_gemv<f16_t, false>(a, st);
    return UAMD_ERR_DTYPE;
}

extern "C" int uamd_gemv(const void* x, int K, const uamd_gemv_group* groups, int n_groups, int nf4, int blocksize,
                         int dtype, void* stream) {
    return gemv_entry(x, K, groups, n_groups, nf4, blocksize, dtype, stream, nullptr);
}

// uamd_gemv with the token PRODUCED inside the launch (one decoder-layer step = 7 launches instead of 14):
//   pro->mode 1: x = SwiGLU(x, x2)                       (fast_swiglu_inference, llama.py:572-606: down_proj's input)
//   pro->mode 2: h = x + res (x may be NULL), x' = rmsnorm(h) * norm_w; block 0 writes h to h_out   (the residual add +
//                fast_rms_layernorm_inference in front of q|k|v, gate|up and lm_head, llama.py:1249-1364)
//   pro->a_rows: t = A x for the stacked LoRA A rows [Rt, K] computed by every block; group g reads t[t_off[g] ..]
//                (the `mv` of fast_linear_forward, utils.py:1107-1117, without a launch of its own)
extern "C" int uamd_gemv_fused(const void* x, int K, const uamd_gemv_group* groups, int n_groups, int nf4, int blocksize,
                               int dtype, void* stream, const uamd_gemv_prologue* pro) {
    return gemv_entry(x, K, groups, n_groups, nf4, blocksize, dtype, stream, pro);
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
        hipLaunchKernelGGL((rope_append_kernel<bf16_t>), dim3(Hq + 2 * Hk, B), dim3(64), 0, st, (bf16_t*)qkv, ld_qkv, (const bf16_t*)cos_t,
                           (const bf16_t*)sin_t, ld_cs, kv_len, rope_pos, (bf16_t*)k_cache, (bf16_t*)v_cache, cache_sb,
                           cache_sh, Hq, Hk, D, s_max);
    else if (dtype == UAMD_F16)
        hipLaunchKernelGGL((rope_append_kernel<f16_t>), dim3(Hq + 2 * Hk, B), dim3(64), 0, st, (f16_t*)qkv, ld_qkv, (const f16_t*)cos_t,
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
    // every group size up to 8 (the kernel loops over its G query heads; Qwen2.5-7B / Qwen2-VL-7B: G = 7)
#define UAMD_DECODE_G(TT)                                                                             \
    switch (G) {                                                                                      \
        case 1: UAMD_DECODE_LAUNCH(TT, 1); break; case 2: UAMD_DECODE_LAUNCH(TT, 2); break;           \
        case 3: UAMD_DECODE_LAUNCH(TT, 3); break; case 4: UAMD_DECODE_LAUNCH(TT, 4); break;           \
        case 5: UAMD_DECODE_LAUNCH(TT, 5); break; case 6: UAMD_DECODE_LAUNCH(TT, 6); break;           \
        case 7: UAMD_DECODE_LAUNCH(TT, 7); break; case 8: UAMD_DECODE_LAUNCH(TT, 8); break;           \
        default: return UAMD_ERR_ARG;                                                                 \
    }
    if (dtype == UAMD_BF16) {
        UAMD_DECODE_G(bf16_t)
    } else if (dtype == UAMD_F16) {
        UAMD_DECODE_G(f16_t)
    } else {
        return UAMD_ERR_DTYPE;
    }
#undef UAMD_DECODE_G
#undef UAMD_DECODE_LAUNCH
    int rc = uamd_launch_status();
    if (rc) return rc;
    if (dtype == UAMD_BF16)
        hipLaunchKernelGGL((attn_decode_combine_kernel<bf16_t>), dim3(Hq, B), dim3(DD), 0, st, partials, (bf16_t*)out, out_sb, Hq, nsplit);
    else
        hipLaunchKernelGGL((attn_decode_combine_kernel<f16_t>), dim3(Hq, B), dim3(DD), 0, st, partials, (f16_t*)out, out_sb, Hq, nsplit);
    return uamd_launch_status();
}
