// 256x256x64 eight-wave free-running MFMA GEMM for gfx950: C[M,N] (+)= A[M,K] @ B[N,K]^T (+ LoRA term), bf16/fp16.
//
// Same contract as the other uamd_gemm_nt* kernels (replaces unsloth/kernels/utils.py:1128-1170 matmul_lora and the
// dX products of unsloth/kernels/fast_lora.py:156,193-204,497-517,639-647).
//
// Structure, from what the two earlier kernels measured:
//   * gemm256.hip (8 waves, two groups in anti-phase through 8 barrier slots per K tile) keeps the matrix pipe
//     60 % busy: every slot lasts as long as the slower of {16 MFMAs, the other group's loads};
//   * gemm_w4.hip (4 waves, one per SIMD, one barrier per tile) loses every cycle its single in-order wave
//     does not overlap, and its 32-deep tiles split each 128-byte line into two L2 requests.
// Here: 8 waves (two per SIMD, so the hardware overlaps one wave's waits with the other's MFMAs), each a
// 128x64 output tile = 4x2 v_mfma_f32_32x32x16 tiles, NO phase barriers: each wave software-pipelines its own
// fragment reads (k-step s+1 into the other register set while the 8 MFMAs of step s issue, MFMA and ds_read
// alternating 1:1), and the block meets at ONE barrier per 64-deep K tile, placed before the first read of the
// next tile. Operands go HBM/L2 -> LDS by LDS-DMA (saddr form), 2 x 64 KiB stages; a wave requests both
// 64-byte halves of a 128-byte line back to back. Bank swizzle and 2-D XCD raster as in gemm_w4.hip.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

template <typename T> struct MfmaF;
template <> struct MfmaF<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct MfmaF<f16_t> {
    typedef f16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int TM = 256, TN = 256, TK = 64, NSTAGE = 2;
constexpr int HALF_BYTES = TM * TK * 2;           // 32 KiB: one operand of one stage
constexpr int STAGE_BYTES = 2 * HALF_BYTES;       // 64 KiB
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;   // 128 KiB
#define UAMD_FR_MAX_GROUPS 3

struct FrArgs {
    const void* A;
    int64_t lda;
    int M, K;
    int n_groups;
    int accumulate;
    int tiles_m;
    int group_m;
    int total_tiles;
    int tile_start[UAMD_FR_MAX_GROUPS + 1];
    uamd_gemm_group g[UAMD_FR_MAX_GROUPS];
};

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// Four LDS-DMA wave-instructions (4 x 1 KiB) in one statement: lane l copies 16 B from base + voff_i to LDS
// [dst_i + 16 l). Inline asm ON PURPOSE (see gemm256.hip): hipcc neither counts nor waits for these, the
// hand-placed counted `s_waitcnt vmcnt(N)` + barrier retire them. M0 is written in the statement that reads it
// and restored (cdna guide 5.7).
__device__ __forceinline__ void dma16x4(const void* base, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                        unsigned d0, unsigned d1, unsigned d2, unsigned d3) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %9\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %9\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %9\n\t"
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %9\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(d0), "s"(d1), "s"(d2), "s"(d3), "s"(base)
        : "memory");
}

template <typename T>
__global__ void __launch_bounds__(512, 2) gemm_nt_fr_kernel(FrArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaF<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // wave owns rows wm*128.., cols wn*64..
    const int l31 = lane & 31, lh = lane >> 5;

    int tile = blockIdx.x;
    {
        const int nt = p.total_tiles, q = nt >> 3, r = nt & 7, x = tile & 7, j = tile >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn_lin;
    {
        const int gm = p.group_m, tiles_n = p.tile_start[UAMD_FR_MAX_GROUPS];
        const int per_group = gm * tiles_n;
        const int grp = tile / per_group;
        const int first_m = grp * gm;
        const int gsz = min(gm, p.tiles_m - first_m);
        const int rem = tile - grp * per_group;
        tn_lin = rem / gsz;
        tm = first_m + (rem - tn_lin * gsz);
    }
    int gi = 0;
#pragma unroll
    for (int i = 1; i < UAMD_FR_MAX_GROUPS; ++i)
        if (i < p.n_groups && tn_lin >= p.tile_start[i]) gi = i;
    const uamd_gemm_group& g = p.g[gi];
    const int m0 = tm * TM, n0 = (tn_lin - p.tile_start[gi]) * TN;
    const int M = p.M, K = p.K, N = g.N;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- DMA plan. A stage holds A as 32 sub-tiles [16 rows x 32 k] of 1 KiB (sub-tile (rg, kh) at
    //      (rg*2+kh)*1024), then B likewise at +32 KiB. Wave w copies row groups 2w, 2w+1, both k halves, of A and
    //      of B: 4 + 4 instructions whose LDS destinations are consecutive. lane -> (row = lane>>2, stored slot =
    //      lane&3); the stored slot holds logical 16-byte k-slot (lane&3) ^ ((row>>2)&3).
    const int sub_row = lane >> 2;
    const int sub_slot = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ra = m0 + (2 * wave + (j >> 1)) * 16 + sub_row;
        int rb = n0 + (2 * wave + (j >> 1)) * 16 + sub_row;
        ra = (ra < M ? ra : M - 1) - m0;          // clamped rows are never stored
        rb = (rb < N ? rb : N - 1) - n0;
        a_off[j] = (unsigned)(((int64_t)ra * p.lda + (j & 1) * 32 + sub_slot * 8) * (int)sizeof(T));
        b_off[j] = (unsigned)(((int64_t)rb * g.ldb + (j & 1) * 32 + sub_slot * 8) * (int)sizeof(T));
    }
    const T* a_tile = (const T*)p.A + (int64_t)m0 * p.lda;      // wave-uniform bases (SGPR pairs)
    const T* b_tile = (const T*)g.B + (int64_t)n0 * g.ldb;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 4096;
    auto issue = [&](int kt, int stage) {
        const unsigned d = dst_w + stage * STAGE_BYTES;
        dma16x4(a_tile + (int64_t)kt * TK, a_off[0], a_off[1], a_off[2], a_off[3], d, d + 1024, d + 2048, d + 3072);
        const unsigned e = d + HALF_BYTES;
        dma16x4(b_tile + (int64_t)kt * TK, b_off[0], b_off[1], b_off[2], b_off[3], e, e + 1024, e + 2048, e + 3072);
    };

    // ---- fragment addresses. 32x32x16 operand: lane -> (row = lane&31, 8 k at (lane>>5)*8) of k-step ks (0..3):
    //      sub-tile k half = ks>>1, logical slot = 2*(ks&1) + (lane>>5), stored slot = logical ^ ((row&15)>>2).
    const int frag_row = (l31 >> 4) * 2048 + (l31 & 15) * 64;
    const int slot0 = lh ^ ((l31 & 15) >> 2);
    const int oA = wm * 16384 + frag_row + slot0 * 16;                       // even k-steps; odd: ^ 32
    const int oB = HALF_BYTES + wn * 8192 + frag_row + slot0 * 16;

    frag_t fa0[4], fb0[2], fa1[4], fb1[2];
#define LDS_FRAG(DST, OFF)                                                                                \
    do {                                                                                                  \
        union { uint4 r; frag_t f; } u_;                                                                  \
        u_.r = *reinterpret_cast<const uint4*>(smem + (OFF));                                             \
        DST = u_.f;                                                                                       \
    } while (0)
    // k-step KS of the stage at byte offset SO into set (FA, FB)
#define READ_SET(FA, FB, SO, KS)                                                                          \
    do {                                                                                                  \
        const int xo_ = ((KS) & 1) * 32, ko_ = ((KS) >> 1) * 1024;                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) LDS_FRAG(FA[i_], (SO) + ((oA ^ xo_) + ko_) + i_ * 4096); \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) LDS_FRAG(FB[j_], (SO) + ((oB ^ xo_) + ko_) + j_ * 4096); \
    } while (0)
#define MMA_SET(FA, FB)                                                                                   \
    do {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                  \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                              \
                acc[i_][j_] = MfmaF<T>::run(FB[j_], FA[i_], acc[i_][j_]);                                 \
    } while (0)
    // 6 ds_read_b128 and the first 6 of 8 MFMAs alternate 1:1; everything is pinned between sched_barriers
#define ALT6()                                                                                            \
    do {                                                                                                  \
        _Pragma("unroll") for (int z_ = 0; z_ < 6; ++z_) {                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                            \
        }                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)

    const int nk = K / TK;     // host guarantees K % 64 == 0
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    READ_SET(fa0, fb0, 0, 0);

    // steady state: every tile but the last has a successor -> no conditions in the loop body
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int so = (kt & 1) * STAGE_BYTES, sn = ((kt + 1) & 1) * STAGE_BYTES;
        issue(kt + 1, (kt + 1) & 1);                 // that stage was drained before the previous barrier
        __builtin_amdgcn_sched_barrier(0);
        READ_SET(fa1, fb1, so, 1); MMA_SET(fa0, fb0); ALT6();
        READ_SET(fa0, fb0, so, 2); MMA_SET(fa1, fb1); ALT6();
        READ_SET(fa1, fb1, so, 3); MMA_SET(fa0, fb0); ALT6();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt+1 landed (this wave's share)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this stage's reads are done: WAR-safe
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        READ_SET(fa0, fb0, sn, 0); MMA_SET(fa1, fb1); ALT6();
    }
    {   // last tile
        const int so = ((nk - 1) & 1) * STAGE_BYTES;
        READ_SET(fa1, fb1, so, 1); MMA_SET(fa0, fb0); ALT6();
        READ_SET(fa0, fb0, so, 2); MMA_SET(fa1, fb1); ALT6();
        READ_SET(fa1, fb1, so, 3); MMA_SET(fa0, fb0); ALT6();
        MMA_SET(fa1, fb1);
    }
#undef ALT6
#undef MMA_SET
#undef READ_SET
#undef LDS_FRAG

    // ---- epilogue (as gemm_w4.hip): lane holds C[m][n..n+3], m = ..+(lane&31), n = ..+8*q + 4*(lane>>5);
    //      LoRA term out = acc + s * (T(XA) @ LB^T) per 32x32 tile right before its store.
    const bool lora = g.lora_xa != nullptr;
    const int R = lora ? g.R : 0;
    const float ls = g.lora_scale;
    T* Cg = (T*)g.C;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 7) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16_t t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
            if (lora) {
                const int nl = n0 + wn * 64 + j * 32 + l31;
                for (int k0 = 0; k0 < R; k0 += 16) {
                    const int k = k0 + lh * 8;
                    uint4 raw = make_uint4(0, 0, 0, 0);
                    if (nl < N && k < R)
                        raw = *reinterpret_cast<const uint4*>((const T*)g.lora_b + (int64_t)nl * g.ld_lb + k);
                    union { uint4 r; frag_t f; } ub; ub.r = raw;
                    Vec16<T> v;
                    v.raw = make_uint4(0, 0, 0, 0);
                    if (m < M && k < R) {
                        const float* src = g.lora_xa + (int64_t)m * g.ld_xa + k;
                        const float4 f0 = *reinterpret_cast<const float4*>(src);
                        const float4 f1 = *reinterpret_cast<const float4*>(src + 4);
                        v.e[0] = from_f32<T>(f0.x); v.e[1] = from_f32<T>(f0.y);
                        v.e[2] = from_f32<T>(f0.z); v.e[3] = from_f32<T>(f0.w);
                        v.e[4] = from_f32<T>(f1.x); v.e[5] = from_f32<T>(f1.y);
                        v.e[6] = from_f32<T>(f1.z); v.e[7] = from_f32<T>(f1.w);
                    }
                    union { uint4 r; frag_t f; } ua; ua.r = v.raw;
                    t = MfmaF<T>::run(ub.f, ua.f, t);
                }
            }
            if (m >= M) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + q * 8 + lh * 4;
                if (n >= N) continue;
                T* dst = Cg + (int64_t)m * g.ldc + n;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q * 4 + r] + ls * t[q * 4 + r];
                if (n + 3 < N && vec_ok) {
                    union { uint2 raw; T e[4]; } o;
                    if (p.accumulate) {
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
                        if (p.accumulate) x += to_f32(dst[r]);
                        dst[r] = from_f32<T>(x);
                    }
                }
            }
        }
    }
}

template <typename T>
int launch_fr(const FrArgs& a, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_fr_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_nt_fr_kernel<T>), dim3((unsigned)a.total_tiles), dim3(512), LDS_BYTES, st, a);
    return uamd_launch_status();
}

}  // namespace

// Same contract as uamd_gemm_nt (dense B), 256x256x64 tiles, 8 free-running waves. Requires K % 64 == 0.
extern "C" int uamd_gemm_nt_fr(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                               int n_groups, int accumulate, int dtype, void* stream) {
    if (M < 0 || K <= 0 || n_groups < 1 || n_groups > UAMD_FR_MAX_GROUPS || !groups) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 63) || (lda & 7) || !aligned16(A)) return UAMD_ERR_ALIGN;
    if (lda > (int64_t)(1 << 22)) return UAMD_ERR_ARG;        // 32-bit per-lane byte offsets inside a tile
    FrArgs a;
    a.A = A; a.lda = lda; a.M = M; a.K = K; a.n_groups = n_groups; a.accumulate = accumulate;
    a.tiles_m = (M + TM - 1) / TM;
    int tn = 0;
    for (int i = 0; i < UAMD_FR_MAX_GROUPS; ++i) {
        a.tile_start[i] = tn;
        if (i < n_groups) {
            const uamd_gemm_group& g = groups[i];
            if (g.N <= 0 || !g.B || !g.C) return UAMD_ERR_ARG;
            if ((g.ldb & 7) || !aligned16(g.B)) return UAMD_ERR_ALIGN;
            if (g.ldb > (int64_t)(1 << 22)) return UAMD_ERR_ARG;
            if (g.lora_xa) {
                if (!g.lora_b || g.R <= 0 || (g.R & 7) || (g.ld_xa & 3) || (g.ld_lb & 7) ||
                    !aligned16(g.lora_xa) || !aligned16(g.lora_b))
                    return UAMD_ERR_ALIGN;
            }
            a.g[i] = g;
            tn += (g.N + TN - 1) / TN;
        } else {
            a.g[i] = groups[0];
        }
    }
    a.tile_start[UAMD_FR_MAX_GROUPS] = tn;
    const int64_t total = (int64_t)tn * a.tiles_m;
    if (total > 0x7fffffffLL) return UAMD_ERR_ARG;
    a.total_tiles = (int)total;
    {
        const int gm = uamd_tuning_get(UAMD_TUNE_GROUP_M);
        a.group_m = gm < 1 ? 1 : (gm < a.tiles_m ? gm : a.tiles_m);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) return launch_fr<bf16_t>(a, st);
    if (dtype == UAMD_F16) return launch_fr<f16_t>(a, st);
    return UAMD_ERR_DTYPE;
}
