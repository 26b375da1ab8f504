// 256x256x32 four-wave MFMA GEMM for gfx950: C[M,N] (+)= A[M,K] @ B[N,K]^T (+ LoRA term), bf16/fp16.
//
// Same contract as gemm_nt_kernel / gemm_nt256_kernel (replaces unsloth/kernels/utils.py:1128-1170 matmul_lora
// and the dX products of unsloth/kernels/fast_lora.py:156,193-204,497-517,639-647). This is the large-M kernel.
//
// Why a third structure. gemm256.hip runs 8 waves (2 per SIMD) in two anti-phase groups: in every barrier slot
// one group feeds the matrix pipe 16 MFMAs (~256 cycles) while the other issues its LDS reads + DMA, and the
// slot lasts as long as the SLOWER of the two (measured ~500 cycles: 8 barriers per 64-deep K tile, ~52 % of
// the MFMA peak). Here the block is 4 waves = ONE wave per SIMD, each owning a 128x128 output tile in the whole
// 512-entry register file (256 accumulator registers + two fragment sets):
//   * v_mfma_f32_32x32x16: 32 cycles per instruction, highest MFMA ceiling on CDNA4, half the instruction count
//     of 16x16x32 for the same flops;
//   * per 32-deep K tile a wave issues 32 MFMAs (1024 cycles of matrix pipe) against 16 ds_read_b128 + 8 LDS-DMA
//     instructions: <= 2 fillers per MFMA gap, so one in-order wave hides them behind its own MFMAs (fragments
//     for k-step s+1 are read into the OTHER register set before the MFMAs of step s are issued);
//   * ONE barrier per K tile (not 8), placed before the reads of the next tile and behind a counted
//     `s_waitcnt vmcnt(16)`: two whole tiles of DMA stay in flight across every barrier (4-stage LDS ring,
//     4 x 32 KiB), so HBM/L2 latency is covered by ~3 tiles = ~3000 cycles of MFMA work;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, saddr form: wave-uniform tile base in
//     SGPRs + one 32-bit per-lane offset), no VGPR round trip, no ds_write. One 1-KiB DMA instruction fills one
//     [16 rows x 32 k] sub-tile; the bank swizzle (16-byte slot ^= (row>>2)&3, conflict-free for the
//     32x32x16 fragment pattern of ds_read_b128) is applied on the per-lane SOURCE address because the DMA
//     destination is lane-linear.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma32<f16_t> {
    typedef f16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int TM = 256, TN = 256, TK = 32, NSTAGE = 4;
constexpr int HALF_BYTES = TM * TK * 2;           // 16 KiB: one operand of one stage
constexpr int STAGE_BYTES = 2 * HALF_BYTES;       // 32 KiB
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;   // 128 KiB
#define UAMD_W4_MAX_GROUPS 3

struct W4Args {
    const void* A;
    int64_t lda;
    int M, K;
    int n_groups;
    int accumulate;
    int tiles_m;
    int group_m;
    int total_tiles;
    int tile_start[UAMD_W4_MAX_GROUPS + 1];
    uamd_gemm_group g[UAMD_W4_MAX_GROUPS];
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

// VAR selects how the K-tile body is handed to hipcc's scheduler (A/B-tested on hardware):
//   0 = source order is a hint only;  1 = 4-MFMA / 4-load chunks pinned with sched_barrier;
//   2 = pinned segments, inside them MFMA and ds_read alternate 1:1 (sched_group_barrier).
template <typename T, int VAR>
__global__ void __launch_bounds__(256, 1) gemm_nt_w4_kernel(W4Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma32<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;          // wave owns rows wm*128.., cols wn*128..
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- tile mapping with an XCD-aware, bijective remap (block b runs on XCD b % 8): each XCD gets a
    //      contiguous run of tiles, so the A/B panels it re-reads stay in ITS 4 MiB L2.
    int tile = blockIdx.x;
    {
        const int nt = p.total_tiles, q = nt >> 3, r = nt & 7, x = tile & 7, j = tile >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    // Grouped raster inside the run: 32 consecutive tiles (= the tiles one XCD's 32 CUs work on together) cover
    // group_m row panels x 32/group_m column panels instead of 32 x 1, so every A panel is shared by 32/group_m
    // co-running tiles and every B panel by group_m: ~2.7x less L2-miss (fabric) traffic than m-fastest order,
    // which measured ~10x the algorithmic bytes (rocprofv3 TCC_MISS / FETCH_SIZE, profiles/r01_gemm_pmc.md).
    int tm, tn_lin;
    {
        const int gm = p.group_m, tiles_n = p.tile_start[UAMD_W4_MAX_GROUPS];
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
    for (int i = 1; i < UAMD_W4_MAX_GROUPS; ++i)
        if (i < p.n_groups && tn_lin >= p.tile_start[i]) gi = i;
    const uamd_gemm_group& g = p.g[gi];
    const int m0 = tm * TM, n0 = (tn_lin - p.tile_start[gi]) * TN;
    const int M = p.M, K = p.K, N = g.N;

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- DMA plan. A stage holds A rows as 16 sub-tiles [16 rows x 32 k] of 1 KiB (sub-tile rg at rg*1024),
    //      then B likewise at +16 KiB. Wave w issues sub-tiles rg = c*4 + w (c = 0..3) of A and of B.
    //      lane -> (row = lane>>2, stored slot = lane&3) inside the sub-tile; the stored slot holds logical
    //      16-byte k-slot (lane&3) ^ ((row>>2)&3).
    const int sub_row = lane >> 2;
    const int sub_slot = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int ra = m0 + (c * 4 + wave) * 16 + sub_row;
        int rb = n0 + (c * 4 + wave) * 16 + sub_row;
        ra = (ra < M ? ra : M - 1) - m0;          // clamped rows are never stored
        rb = (rb < N ? rb : N - 1) - n0;
        a_off[c] = (unsigned)(((int64_t)ra * p.lda + sub_slot * 8) * (int)sizeof(T));
        b_off[c] = (unsigned)(((int64_t)rb * g.ldb + sub_slot * 8) * (int)sizeof(T));
    }
    const T* a_tile = (const T*)p.A + (int64_t)m0 * p.lda;      // wave-uniform bases (SGPR pairs)
    const T* b_tile = (const T*)g.B + (int64_t)n0 * g.ldb;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 1024;
    // 4 + 4 DMA wave-instructions: this wave's share of tile kt
    auto issue_a = [&](int kt, int stage) {
        const unsigned d = dst_w + stage * STAGE_BYTES;
        dma16x4(a_tile + (int64_t)kt * TK, a_off[0], a_off[1], a_off[2], a_off[3], d, d + 4096, d + 8192, d + 12288);
    };
    auto issue_b = [&](int kt, int stage) {
        const unsigned e = dst_w + stage * STAGE_BYTES + HALF_BYTES;
        dma16x4(b_tile + (int64_t)kt * TK, b_off[0], b_off[1], b_off[2], b_off[3], e, e + 4096, e + 8192, e + 12288);
    };

    // ---- fragment addresses. 32x32x16 operand: lane -> (row = lane&31, 8 k at (lane>>5)*8) of k-step ks;
    //      logical slot = 2*ks + (lane>>5), stored slot = logical ^ ((row&15)>>2).
    const int frag_row = (l31 >> 4) * 1024 + (l31 & 15) * 64;
    const int slot0 = lh ^ ((l31 & 15) >> 2);
    const int offA0 = wm * 8192 + frag_row + slot0 * 16;                 // k-step 0
    const int offA1 = wm * 8192 + frag_row + (slot0 ^ 2) * 16;           // k-step 1
    const int offB0 = HALF_BYTES + wn * 8192 + frag_row + slot0 * 16;
    const int offB1 = HALF_BYTES + wn * 8192 + frag_row + (slot0 ^ 2) * 16;

    frag_t fa0[4], fb0[4], fa1[4], fb1[4];
#define LDS_FRAG(DST, OFF)                                                                                \
    do {                                                                                                  \
        union { uint4 r; frag_t f; } u_;                                                                  \
        u_.r = *reinterpret_cast<const uint4*>(smem + (OFF));                                             \
        DST = u_.f;                                                                                       \
    } while (0)
#define READ_A(FA, OA)                                                                                    \
    do { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) LDS_FRAG(FA[i_], (OA) + i_ * 2048); } while (0)
#define READ_B(FB, OB)                                                                                    \
    do { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) LDS_FRAG(FB[j_], (OB) + j_ * 2048); } while (0)
#define MMA_ROW(FA, FB, I)                                                                                \
    do { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[I][j_] = Mfma32<T>::run(FB[j_], FA[I], acc[I][j_]); } while (0)

    const int nk = K / TK;     // host guarantees K % 32 == 0
    // ---- prologue: tiles 0..2 in flight, tile 0 landed, its first fragment set read
    issue_a(0, 0); issue_b(0, 0);
    if (nk > 1) { issue_a(1, 1); issue_b(1, 1); }
    if (nk > 2) { issue_a(2, 2); issue_b(2, 2); }
    if (nk > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    READ_A(fa0, offA0); READ_B(fb0, offB0);

    // One K tile (source order = intended issue order; MFMAs are register-only, so hipcc may slide them between
    // the loads, which is what we want):
    //   [DMA A of tile kt+3 -> the stage drained one barrier ago] 4 MFMA | [read A k-step 1 -> set 1] 4 MFMA |
    //   [DMA B of tile kt+3] 4 MFMA | [read B k-step 1 -> set 1] 4 MFMA |
    //   [wait: tile kt+1 landed, two later tiles may stay in flight] [barrier]
    //   [read A, B k-step 0 of tile kt+1 -> set 0] 16 MFMA on set 1
    // STEADY = the tile has three successors: no conditions in the body.
#define PIN() do { if (VAR >= 1) __builtin_amdgcn_sched_barrier(0); } while (0)
#define ALT8()                                                                                            \
    do {                                                                                                  \
        if (VAR == 2) {                                                                                   \
            _Pragma("unroll") for (int z_ = 0; z_ < 8; ++z_) {                                            \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
#define TILE_BODY(STEADY)                                                                                 \
    do {                                                                                                  \
        const int so_ = (kt & 3) * STAGE_BYTES, sn_ = ((kt + 1) & 3) * STAGE_BYTES;                       \
        const bool more3_ = (STEADY) || (kt + 3 < nk);                                                    \
        if (more3_) issue_a(kt + 3, (kt + 3) & 3);                                                        \
        PIN();                                                                                            \
        if (VAR == 2) {                                                                                   \
            READ_A(fa1, so_ + offA1); READ_B(fb1, so_ + offB1);                                           \
            MMA_ROW(fa0, fb0, 0); MMA_ROW(fa0, fb0, 1);                                                   \
            ALT8(); PIN();                                                                                \
            if (more3_) issue_b(kt + 3, (kt + 3) & 3);                                                    \
            PIN();                                                                                        \
            MMA_ROW(fa0, fb0, 2); MMA_ROW(fa0, fb0, 3);                                                   \
            PIN();                                                                                        \
        } else {                                                                                          \
            MMA_ROW(fa0, fb0, 0); PIN();                                                                  \
            READ_A(fa1, so_ + offA1); PIN();                                                              \
            MMA_ROW(fa0, fb0, 1); PIN();                                                                  \
            if (more3_) issue_b(kt + 3, (kt + 3) & 3);                                                    \
            PIN();                                                                                        \
            MMA_ROW(fa0, fb0, 2); PIN();                                                                  \
            READ_B(fb1, so_ + offB1); PIN();                                                              \
            MMA_ROW(fa0, fb0, 3); PIN();                                                                  \
        }                                                                                                 \
        if ((STEADY) || kt + 1 < nk) {                                                                    \
            if (more3_) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                 \
            else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                        \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* this stage's reads are done: WAR-safe */ \
            __builtin_amdgcn_s_barrier();                                                                 \
            asm volatile("" ::: "memory");                                                                \
            PIN();                                                                                        \
            if (VAR == 2) {                                                                               \
                READ_A(fa0, sn_ + offA0); READ_B(fb0, sn_ + offB0);                                       \
                MMA_ROW(fa1, fb1, 0); MMA_ROW(fa1, fb1, 1);                                               \
                ALT8(); PIN();                                                                            \
            } else {                                                                                      \
                READ_A(fa0, sn_ + offA0); PIN();                                                          \
                MMA_ROW(fa1, fb1, 0); PIN();                                                              \
                READ_B(fb0, sn_ + offB0); PIN();                                                          \
                MMA_ROW(fa1, fb1, 1); PIN();                                                              \
            }                                                                                             \
        } else {                                                                                          \
            MMA_ROW(fa1, fb1, 0);                                                                         \
            MMA_ROW(fa1, fb1, 1);                                                                         \
        }                                                                                                 \
        MMA_ROW(fa1, fb1, 2);                                                                             \
        MMA_ROW(fa1, fb1, 3);                                                                             \
        PIN();                                                                                            \
    } while (0)

    int kt = 0;
    for (; kt + 3 < nk; ++kt) TILE_BODY(true);
    for (; kt < nk; ++kt) TILE_BODY(false);
#undef TILE_BODY
#undef PIN
#undef ALT8
#undef READ_A
#undef READ_B
#undef MMA_ROW
#undef LDS_FRAG

    // ---- epilogue. Operands were passed swapped (B rows as the first MFMA operand), so lane holds
    //      C[m][n..n+3] with m = ..+(lane&31) and n = ..+8*q + 4*(lane>>5) for register quad q.
    //      LoRA term (utils.py:1162-1168): out = acc + s * (T(XA) @ LB^T), one 32x32x16 k-step per 16 ranks,
    //      computed per 32x32 tile right before its store (no 256-register scaling pass over acc).
    const bool lora = g.lora_xa != nullptr;
    const int R = lora ? g.R : 0;
    const float ls = g.lora_scale;
    T* Cg = (T*)g.C;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 7) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x16_t t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
            if (lora) {
                const int nl = n0 + wn * 128 + j * 32 + l31;
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
                    t = Mfma32<T>::run(ub.f, ua.f, t);
                }
            }
            if (m >= M) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 128 + j * 32 + q * 8 + lh * 4;
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

template <typename T, int VAR>
int launch_w4v(const W4Args& a, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_w4_kernel<T, VAR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_nt_w4_kernel<T, VAR>), dim3((unsigned)a.total_tiles), dim3(256), LDS_BYTES, st, a);
    return uamd_launch_status();
}

template <typename T>
int launch_w4(const W4Args& a, hipStream_t st, int variant) {
    switch (variant) {
        case 0: return launch_w4v<T, 0>(a, st);
        case 2: return launch_w4v<T, 2>(a, st);
        default: return launch_w4v<T, 1>(a, st);
    }
}

}  // namespace

// Same contract as uamd_gemm_nt (dense B), 256x256x32 tiles, 4 waves. Requires K % 32 == 0.
extern "C" int uamd_gemm_nt_w4(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                               int n_groups, int accumulate, int dtype, void* stream) {
    if (M < 0 || K <= 0 || n_groups < 1 || n_groups > UAMD_W4_MAX_GROUPS || !groups) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 31) || (lda & 7) || !aligned16(A)) return UAMD_ERR_ALIGN;
    if (lda > (int64_t)(1 << 22)) return UAMD_ERR_ARG;        // 32-bit per-lane byte offsets inside a tile
    W4Args a;
    a.A = A; a.lda = lda; a.M = M; a.K = K; a.n_groups = n_groups; a.accumulate = accumulate;
    a.tiles_m = (M + TM - 1) / TM;
    int tn = 0;
    for (int i = 0; i < UAMD_W4_MAX_GROUPS; ++i) {
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
    a.tile_start[UAMD_W4_MAX_GROUPS] = tn;
    const int64_t total = (int64_t)tn * a.tiles_m;
    if (total > 0x7fffffffLL) return UAMD_ERR_ARG;
    a.total_tiles = (int)total;
    {
        const int gm = uamd_tuning_get(UAMD_TUNE_GROUP_M);
        a.group_m = gm < 1 ? 1 : (gm < a.tiles_m ? gm : a.tiles_m);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) return launch_w4<bf16_t>(a, st, uamd_tuning_get(UAMD_TUNE_W4_VARIANT));
    if (dtype == UAMD_F16) return launch_w4<f16_t>(a, st, uamd_tuning_get(UAMD_TUNE_W4_VARIANT));
    return UAMD_ERR_DTYPE;
}
