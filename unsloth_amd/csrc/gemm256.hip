// 256x256x64 "ping-pong" MFMA GEMM for gfx950: C[M,N] (+)= A[M,K] @ B[N,K]^T (+ LoRA term), bf16/fp16.
//
// Same contract as gemm_nt_kernel (gemm.hip; replaces unsloth/kernels/utils.py:1128-1170 matmul_lora and the
// dX products of unsloth/kernels/fast_lora.py), built for large M (tokens >= 4096) where a 256x256 tile still
// gives >= 2 tiles per CU. Why a second kernel: the 128x128 register-staged kernel is LDS-bound on CDNA4 --
// every operand byte crosses the VGPR->LDS store path (ds_write_b128 = 13 cycles / KiB) -- and tops out near
// 0.85 PFLOP/s. Here:
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write;
//   * one 1-KiB DMA instruction fills [8 rows x 64 k] = 8 full 128-byte lines (half of a 16-row MFMA tile, all of
//     TK); the bank swizzle (16-byte slot ^= (row>>1)&7, conflict-free for ds_read_b128) is applied on the
//     per-lane SOURCE address because the DMA destination is lane-linear;
//   * 8 waves = 2 groups of 4 (group = wave>>2 owns 128 rows). The groups run in ANTI-PHASE: in every
//     barrier-delimited slot one group issues its LDS fragment reads + next tiles' DMA while the other group
//     owns the matrix pipe for 16 back-to-back MFMAs (two waves share a SIMD: one computes, one loads);
//   * K tiles are double-buffered in LDS (2 x 64 KiB); a wave's 8 DMA pieces of tile t+1 are issued 3/3/2
//     over three load slots starting as soon as tile t-1's buffer is drained, and retired by ONE counted
//     `s_waitcnt vmcnt(3)` per tile (never a full drain in the loop).
// Slot timeline for tile t (group 1 is one slot behind):
//   L0: read A[rows 0..63] (8 frags) + B[cols 0..31] (4) | DMA pieces 3,4,5 of tile t+1 | lgkmcnt(0) | barrier
//   M0: 16 MFMA (quadrant 0,0)                                                                     | barrier
//   L1: read B[cols 32..63] (4)                          | DMA pieces 6,7 of tile t+1   | lgkmcnt(0) | barrier
//   M1: 16 MFMA (0,1)                                                                               | barrier
//   L2: read A[rows 64..127] (8)                                                        | lgkmcnt(0) | barrier
//   M2: 16 MFMA (1,1)                                                                               | barrier
//   L3: DMA pieces 0,1,2 of tile t+2 (buffer of tile t is drained) | vmcnt(3): tile t+1 landed      | barrier
//   M3: 16 MFMA (1,0)                                                                               | barrier
// The LoRA term s*(X A^T) B^T is NOT a separate prologue: the rank block rides the same pipeline as Rk/64 EXTRA K
// TILES whose DMA sources are XK = T(X A^T) [M, Rk] and BK = T(s B) [N, Rk] (zero-padded to 64 columns, see
// uamd_gemm_group.lora_xk). One extra tile per output tile costs 64/K of the launch (1.6 % at K = 4096, 0.4 % at
// K = 14336); the round-1 register prologue (fp32 XA loads + conversions + 32 MFMAs before the loop) cost 4-8 %.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

template <typename T> struct Mfma2;
template <> struct Mfma2<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma2<f16_t> {
    typedef f16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int STAGE_BYTES = (TM + TN) * TK * 2;   // 64 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;        // 128 KiB
#define UAMD_G256_MAX_GROUPS 3

struct G256Args {
    const void* A;
    int64_t lda;
    int M, K;
    int n_groups;
    int accumulate;
    int tiles_m;
    int group_m;
    int total_tiles;
    int tile_start[UAMD_G256_MAX_GROUPS + 1];
    uamd_gemm_group g[UAMD_G256_MAX_GROUPS];
};

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;

// One wave-instruction of LDS-DMA: 64 lanes x 16 B from per-lane global addresses -> LDS [lds_addr,
// lds_addr + 1024). Issued through inline asm ON PURPOSE: with the builtin, hipcc treats every later ds_read
// as possibly aliasing the in-flight DMA and emits `s_waitcnt vmcnt(0)` in front of each fragment-read group,
// which serialises the pipeline. In asm the compiler neither counts nor waits for it; completion is tracked
// by the hand-placed counted `s_waitcnt vmcnt(N)` + barrier (cdna guide 5.7: M0 is written in the same
// statement that reads it, and restored).
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(lds_addr)
        : "memory");
}

// The same with the address split as SGPR base (64-bit, wave-uniform: matrix base + K-tile offset, advanced on the
// scalar unit) + per-lane 32-bit byte offset (row * ld + swizzled slot, fixed for the whole kernel): no 64-bit
// VALU add per piece and half the address registers.
__device__ __forceinline__ void dma16s(unsigned voff, uint64_t sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}

#define SLOT_BARRIER()                          \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

// -DUAMD_G256_TRACE: s_memtime stamps at slot boundaries of K tiles 16 and 17 (tools/gemm_trace.py); never in
// the shipped library.
#ifdef UAMD_G256_TRACE
__device__ unsigned* g_trace256 = nullptr;
#endif
#if defined(UAMD_G256_TRACE) && UAMD_G256_TRACE == 1
#define STAMP(KT, I)                                                                    \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if ((KT) == 16 || (KT) == 17) {                                                 \
            ts[(((KT) & 1) << 4) + (I)] = (unsigned)__builtin_amdgcn_s_memtime();       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        }                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else
#define STAMP(KT, I) do { } while (0)
#endif

// BNN = false: B_g is [N, K] (K contiguous, the forward's weight layout): C = A @ B^T.
// BNN = true : B_g is [K, N] (N contiguous): C = A @ B -- the dX products contract over the weight's ROWS, so they
//              read the SAME row-major decode as the forward and no transposed copy of W is ever written. The B
//              tile is then [64 k][256 n] in LDS (DMA pieces of 2 k-rows x 512 B), and the MFMA operand -- 8
//              consecutive k for one n per lane -- comes out of it through two ds_read_b64_tr_b16 (each hands a
//              lane 4 consecutive k of its column; semantics probed in profiles/r01_tr_probe.txt). Bank swizzle:
//              32-byte granule ^= (k & 3) | ((k >> 3) & 1) << 2, applied on the DMA source column: the 8 rows a
//              32-lane half of the transposing read touches land in 8 different granules of the 256-byte bank row.
// ATN = true (with BNN): A is [K, M] (M contiguous, lda = row stride) as well -- C = A^T @ B, the weight-gradient
//              product dW[out, in] = dY[T, out]^T @ X[T, in] of full fine-tuning: BOTH operands are read in the layout
//              the forward / backward left them in, the A tile goes through the same [64 k][256 m] LDS image and
//              transposing reads as B's. No rank block in this form.
template <typename T, bool BNN, bool ATN = false>
__global__ void __launch_bounds__(512, 2) gemm_nt256_kernel(G256Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma2<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;      // group owns rows grp*128.., wave owns cols wn*64..
    const int l15 = lane & 15, l4 = lane >> 4;

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
        const int gm = p.group_m, tiles_n = p.tile_start[UAMD_G256_MAX_GROUPS];
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
    for (int i = 1; i < UAMD_G256_MAX_GROUPS; ++i)
        if (i < p.n_groups && tn_lin >= p.tile_start[i]) gi = i;
    const uamd_gemm_group& g = p.g[gi];
    const int m0 = tm * TM, n0 = (tn_lin - p.tile_start[gi]) * TN;
    const int M = p.M, K = p.K, N = g.N;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- DMA source pointers. Piece c (0..7) of a tile, issued by wave w, fills sub-tile u = c*8 + w =
    //      [8 rows x 64 k] = 8 FULL 128-byte lines (u < 32: A rows u*8.., u >= 32: B rows (u-32)*8..). Full lines
    //      matter: the L2 serves requests, not bytes -- [16 rows x 64 B] half-line pieces deliver 36 B/clk/CU with
    //      all CUs streaming, full-line pieces 59 (tools/probes/dma_probe.hip, profiles/r01_dma_probe.txt).
    //      lane -> (row = lane>>3, 16-byte slot position q = lane&7) in LDS (the DMA destination is lane-linear);
    //      the bank swizzle is applied on the SOURCE: position q of row r holds k-slot q ^ f(r), f = (r16>>1)&7
    //      with r16 the row inside its 16-row MFMA tile (conflict-free for all four ds_read_b128 lane groups).
    const int sub_row = lane >> 3;
    const int sub_slot = (lane & 7) ^ (((wave & 1) << 2) | (sub_row >> 1));
    auto sgpr64 = [](const void* q) {               // a wave-uniform pointer as a value the compiler keeps in SGPRs
        const uint64_t u = (uint64_t)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    const uint64_t a_gbase = sgpr64(p.A), b_gbase = sgpr64(g.B);
    unsigned a_off[4], b_off[4];                    // byte offsets of this lane's row / slot (host: < 4 GiB)
    // BNN: lane -> (k-row of the piece's pair = lane >> 5, physical 16-byte slot = lane & 31 of the 512-byte row)
    const int nn_krow = lane >> 5;
    auto nn_col = [&](int krow) {                   // source column (elements) of this lane's slot in k-row `krow`
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        int col = n0 + (((lane & 31) ^ (f << 1)) << 3);
        return col + 8 <= N ? col : N - 8;          // columns past N are never stored
    };
    auto tn_col = [&](int krow) {                   // the same for A [K, M]
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        int col = m0 + (((lane & 31) ^ (f << 1)) << 3);
        return col + 8 <= M ? col : M - 8;
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (ATN) {
            const int krow = (c * 8 + wave) * 2 + nn_krow;
            a_off[c] = (unsigned)(((int64_t)krow * p.lda + tn_col(krow)) * (int64_t)sizeof(T));
        } else {
            int ra = m0 + (c * 8 + wave) * 8 + sub_row;
            ra = ra < M ? ra : M - 1;          // clamped rows are never stored
            a_off[c] = (unsigned)(((int64_t)ra * p.lda + sub_slot * 8) * (int64_t)sizeof(T));
        }
        if (BNN) {
            const int krow = (c * 8 + wave) * 2 + nn_krow;
            b_off[c] = (unsigned)(((int64_t)krow * g.ldb + nn_col(krow)) * (int64_t)sizeof(T));
        } else {
            int rb = n0 + (c * 8 + wave) * 8 + sub_row;
            rb = rb < N ? rb : N - 1;
            b_off[c] = (unsigned)(((int64_t)rb * g.ldb + sub_slot * 8) * (int64_t)sizeof(T));
        }
    }
    // bytes from one K tile of B to the next: 64 columns (NT) or 64 rows (NN)
    const uint64_t b_tile_step = BNN ? (uint64_t)__builtin_amdgcn_readfirstlane((int)g.ldb) * (TK * sizeof(T))
                                     : (uint64_t)(TK * sizeof(T));
    const uint64_t a_tile_step = ATN ? (uint64_t)__builtin_amdgcn_readfirstlane((int)p.lda) * (TK * sizeof(T))
                                     : (uint64_t)(TK * sizeof(T));
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;    // LDS byte address of the dynamic region
    // K tiles: nk_main of the operands proper, then the rank block's. Both counts are wave-uniform by construction;
    // readfirstlane tells the compiler so (loop bounds and branches on them stay on the scalar unit).
    const int nk_main = __builtin_amdgcn_readfirstlane(K / TK);      // host guarantees K % 64 == 0
    const int nk = __builtin_amdgcn_readfirstlane(nk_main + (g.lora_xk != nullptr ? g.Rk / TK : 0));
    // issue_main: the hot path, branch-free (tiles of A / B proper). c, stage are compile-time at every call site.
    auto issue_main = [&](int c, int kt, int stage) {
        const unsigned dst = lds_base + stage * STAGE_BYTES + (c * 8 + wave) * 1024;
        if (c < 4) dma16s(a_off[c & 3], a_gbase + (uint64_t)kt * a_tile_step, dst);
        else dma16s(b_off[c & 3], b_gbase + (uint64_t)kt * b_tile_step, dst);
    };
    // issue_any: used only by the prologue and the last few tiles, where the tile being fetched may belong to the
    // rank block: same piece geometry, sources are XK / BK rows (addresses rebuilt here: no registers are held
    // for them during the main loop)
    // The rank block's base pointers / strides are pulled into SGPRs HERE, once (readfirstlane makes them computed
    // values, not re-loadable kernel arguments: with plain reads the compiler sank an s_load + s_waitcnt lgkmcnt(0)
    // pair -- which also drains every LDS read in flight -- into each of the 11 rank-tile DMA sites of a tile).
    const uint64_t xk_base = sgpr64(g.lora_xk), bk_base = sgpr64(g.lora_bk);
    const int ld_xk = __builtin_amdgcn_readfirstlane((int)g.ld_xk), ld_bk = __builtin_amdgcn_readfirstlane((int)g.ld_bk);
    auto issue_any = [&](int c, int kt, int stage) {
        if (kt < nk_main) {
            issue_main(c, kt, stage);
        } else {
            const unsigned dst = lds_base + stage * STAGE_BYTES + (c * 8 + wave) * 1024;
            if (BNN && c >= 4) {                         // BK is [Rk, N] here: same geometry as B proper
                const int krow = ((c & 3) * 8 + wave) * 2 + nn_krow;
                const unsigned off = (unsigned)(((int64_t)krow * ld_bk + nn_col(krow)) * (int64_t)sizeof(T));
                dma16s(off, bk_base + (uint64_t)(kt - nk_main) * ((uint64_t)ld_bk * (TK * sizeof(T))), dst);
            } else {
                int row = (c < 4 ? m0 : n0) + ((c & 3) * 8 + wave) * 8 + sub_row;
                const int last = (c < 4 ? M : N) - 1;
                row = row < last ? row : last;
                const int ld = c < 4 ? ld_xk : ld_bk;
                const unsigned off = (unsigned)(((int64_t)row * ld + sub_slot * 8) * (int64_t)sizeof(T));
                dma16s(off, (c < 4 ? xk_base : bk_base) + (uint64_t)(kt - nk_main) * (TK * sizeof(T)), dst);
            }
        }
    };

    // ---- fragment read offsets (bytes inside a stage): 16-row tile i of A at i*2 KiB ([16 rows][128 B]),
    //      B tiles after the 32 KiB of A; lane reads row l15, k-slot (ks*4 + l4) ^ f(l15)
    const int frag_off0 = l15 * 128 + ((l4 ^ ((l15 >> 1) & 7)) << 4);
    const int frag_off[2] = {frag_off0, frag_off0 ^ 64};
    const int a_base = (grp * 8) * 2048;
    const int b_base = 32 * 1024 + (wn * 4) * 2048;
    frag_t af[4][2], bf[4][2];     // A: 4 m-tiles of the current 64-row half x 2 k-halves; B: 4 n-tiles x 2
    // BNN fragment addresses: k-row ks*32 + l4*8 + (l15 >> 2) (+4 for the second read), logical granule wn*4 + t
    const int nn_f = (l15 >> 2) | ((l4 & 1) << 2);
    const int nn_lane = 32 * 1024 + (l4 * 8 + (l15 >> 2)) * 512 + (l15 & 3) * 8;
    auto read_a = [&](int stage, int mq) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ATN) {                       // [64 k][256 m] image at the start of the stage: m-tile grp*8 + mq*4 + i
                    union { s16x4_t h[2]; frag_t f; } u;
                    const int a0 = stage * STAGE_BYTES + (nn_lane - 32 * 1024) + ks * (32 * 512) + (((grp * 8 + mq * 4 + i) ^ nn_f) << 5);
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + a0));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + a0 + 4 * 512));
                    af[i][ks] = u.f;
                } else {
                    union { uint4 r; frag_t f; } u;
                    u.r = *reinterpret_cast<const uint4*>(smem + stage * STAGE_BYTES + a_base + (mq * 4 + i) * 2048 + frag_off[ks]);
                    af[i][ks] = u.f;
                }
            }
    };
    auto read_b = [&](int stage, int nq) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BNN) {
                    union { s16x4_t h[2]; frag_t f; } u;
                    const int a0 = stage * STAGE_BYTES + nn_lane + ks * (32 * 512) + (((wn * 4 + nq * 2 + j) ^ nn_f) << 5);
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + a0));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + a0 + 4 * 512));
                    bf[nq * 2 + j][ks] = u.f;
                } else {
                    union { uint4 r; frag_t f; } u;
                    u.r = *reinterpret_cast<const uint4*>(smem + stage * STAGE_BYTES + b_base + (nq * 2 + j) * 2048 + frag_off[ks]);
                    bf[nq * 2 + j][ks] = u.f;
                }
            }
    };
    auto mma = [&](int mq, int nq) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mq * 4 + i][nq * 2 + j] = Mfma2<T>::run(bf[nq * 2 + j][ks], af[i][ks], acc[mq * 4 + i][nq * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
#ifdef UAMD_G256_TRACE
    unsigned ts[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) ts[i] = 0;
    ts[30] = (unsigned)__builtin_amdgcn_s_memtime();      // whole-tile cycles (UAMD_G256_TRACE=2: only these two)
#endif
    // ---- prologue: tile 0 completely, first 3 pieces of tile 1
#pragma unroll
    for (int c = 0; c < 8; ++c) issue_main(c, 0, 0);
    if (nk > 1) { issue_any(0, 1, 1); issue_any(1, 1, 1); issue_any(2, 1, 1); }
    __builtin_amdgcn_sched_barrier(0);

    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLOT_BARRIER();
    if (grp == 1) SLOT_BARRIER();          // anti-phase: group 1 runs one slot behind

// ISSUE: issue_main (fast loop: every tile it prefetches is a tile of A / B proper and exists, CHECK = 0) or
// issue_any (last tiles: the prefetched tile may be a rank-block tile or past the end, CHECK = 1)
#define TILE_BODY(STAGE, KT, ISSUE, CHECK)                                                \
    do {                                                                                 \
        /* L0 */                                                                         \
        read_a(STAGE, 0); read_b(STAGE, 0);                                              \
        if (!(CHECK) || (KT) + 1 < nk) { ISSUE(3, (KT) + 1, (STAGE) ^ 1); ISSUE(4, (KT) + 1, (STAGE) ^ 1); ISSUE(5, (KT) + 1, (STAGE) ^ 1); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        STAMP(KT, 8); SLOT_BARRIER(); STAMP(KT, 0);                                      \
        mma(0, 0); STAMP(KT, 9); SLOT_BARRIER(); STAMP(KT, 1);                                                       \
        /* L1 */                                                                         \
        read_b(STAGE, 1);                                                                \
        if (!(CHECK) || (KT) + 1 < nk) { ISSUE(6, (KT) + 1, (STAGE) ^ 1); ISSUE(7, (KT) + 1, (STAGE) ^ 1); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        STAMP(KT, 10); SLOT_BARRIER(); STAMP(KT, 2);                                     \
        mma(0, 1); STAMP(KT, 11); SLOT_BARRIER(); STAMP(KT, 3);                                                       \
        /* L2 */                                                                         \
        read_a(STAGE, 1);                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        STAMP(KT, 12); SLOT_BARRIER(); STAMP(KT, 4);                                     \
        mma(1, 1); STAMP(KT, 13); SLOT_BARRIER(); STAMP(KT, 5);                                                       \
        /* L3: this stage is drained by BOTH groups (their last reads ended >= 1 barrier ago) */ \
        if (!(CHECK) || (KT) + 2 < nk) {                                                 \
            ISSUE(0, (KT) + 2, STAGE); ISSUE(1, (KT) + 2, STAGE); ISSUE(2, (KT) + 2, STAGE); \
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                             \
        } else {                                                                         \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
        }                                                                                \
        STAMP(KT, 14); SLOT_BARRIER(); STAMP(KT, 6);                                     \
        mma(1, 0); STAMP(KT, 15); SLOT_BARRIER(); STAMP(KT, 7);                                                       \
    } while (0)

    int kt = 0;
    // fast loop: tiles kt <= nk_main - 3 prefetch only tiles kt + 1, kt + 2 <= nk_main - 1
    for (; kt + 1 < nk_main - 2; kt += 2) {
        TILE_BODY(0, kt, issue_main, 0);
        TILE_BODY(1, kt + 1, issue_main, 0);
    }
    // the last two or three tiles of A / B and the rank block's tiles (kt is even here)
    for (; kt + 1 < nk; kt += 2) {
        TILE_BODY(0, kt, issue_any, 1);
        TILE_BODY(1, kt + 1, issue_any, 1);
    }
    if (kt < nk) TILE_BODY(0, kt, issue_any, 1);
#undef TILE_BODY

    if (grp == 0) SLOT_BARRIER();          // match group 1's extra barrier
#ifdef UAMD_G256_TRACE
    ts[31] = (unsigned)__builtin_amdgcn_s_memtime();
    if (g_trace256 && lane == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 32; ++i) g_trace256[(blockIdx.x * 8 + wave) * 32 + i] = ts[i];
    }
#endif

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+l15, n = ..+4*l4 (operands were passed swapped)
    T* Cg = (T*)g.C;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 7) == 0);
    const T* bias = (const T*)g.bias;
    auto epi = [&](auto acc_c, auto bias_c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + grp * 128 + i * 16 + l15;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + l4 * 4;
                if (n >= N) continue;
                store_c4<T, decltype(acc_c)::value, decltype(bias_c)::value>(
                    Cg + (int64_t)m * g.ldc + n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], n, N, vec_ok, bias);
            }
        }
    };
    UAMD_EPILOGUE_DISPATCH(epi, p.accumulate, bias);
}

// ------------------------------------------------------------------------------------------------------------
// Persistent form of gemm_nt256_kernel: one block per CU walks the virtual block ids v = blockIdx.x, + gridDim.x, ...
// (same XCD-aware raster as above, so the set of tiles in flight at any time is the one the hardware dispatcher
// would have picked) and treats the K tiles of consecutive OUTPUT tiles as one stream: while the last two K tiles
// of an output tile are multiplied, the DMA slots that would sit idle fetch K tiles 0 and 1 of the next one, so a
// new output tile starts with its operands already in LDS instead of a cold 64-KiB prologue (all 256 CUs bursting
// at once: ~3 us of a ~100 us launch per tile), and the epilogue's stores overlap the next tile's loads. The two
// wave groups keep their one-slot stagger across the boundary; the barrier sequence is the steady-state one.
// Stage parity: an output tile with an odd number of K tiles leaves the stream in stage 1; the tile-local code
// always numbers its stages 0, 1, so the two sets of LDS addresses (DMA destination base, fragment read offsets)
// are held in registers and SWAPPED at such a boundary -- no address arithmetic enters the K loop.
// Host contract (gemm256_entry): K >= 4 * 64, gridDim.x <= total_tiles.
// PLAIN = no accumulate, no bias, known at LAUNCH: that instance's epilogue contains no global load at all. With the
// run-time dispatch inside the kernel (PLAIN = false) the accumulate / bias paths' loads sit on the walk's back-edge, hipcc's
// waitcnt pass cannot prove that none of them is still pending when the next output tile's K loop overwrites their registers,
// and it puts `s_waitcnt vmcnt(0)` into the K LOOP'S HEADER -- executed every iteration, draining the three DMA pieces the
// counted vmcnt(3) exists to keep in flight across the barrier (ISA: vmcnt sequence 0, 3, 3 per iteration; PLAIN: 3, 3).
template <typename T, bool BNN, bool PLAIN>
__global__ void __launch_bounds__(512, 2) gemm_nt256p_kernel(G256Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma2<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int M = p.M, total = p.total_tiles;
    const int nk_main = __builtin_amdgcn_readfirstlane(p.K / TK);

    // virtual block id -> (first row, first column inside its group, group): see gemm_nt256_kernel
    auto decode = [&](int v, int& m0, int& n0, int& gi) {
        int tile;
        {
            const int q = total >> 3, r = total & 7, x = v & 7, j = v >> 3;
            tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        }
        const int gm = p.group_m, tiles_n = p.tile_start[UAMD_G256_MAX_GROUPS];
        const int per_group = gm * tiles_n;
        const int rg = tile / per_group;
        const int first_m = rg * gm;
        const int gsz = min(gm, p.tiles_m - first_m);
        const int rem = tile - rg * per_group;
        const int tn_lin = rem / gsz;
        const int tm = first_m + (rem - tn_lin * gsz);
        int g_ = 0, start = 0;
#pragma unroll
        for (int i = 1; i < UAMD_G256_MAX_GROUPS; ++i)
            if (i < p.n_groups && tn_lin >= p.tile_start[i]) { g_ = i; start = p.tile_start[i]; }
        gi = __builtin_amdgcn_readfirstlane(g_);
        m0 = __builtin_amdgcn_readfirstlane(tm * TM);
        n0 = __builtin_amdgcn_readfirstlane((tn_lin - start) * TN);
    };
    auto rank_tiles = [&](int gi) {
        const uamd_gemm_group& g = p.g[gi];
        return __builtin_amdgcn_readfirstlane(g.lora_xk != nullptr ? g.Rk / TK : 0);
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- the ISSUE context: where DMA pieces come from. It belongs to the output tile being multiplied until its
    //      last pieces are issued (L1 of its second-to-last K tile), then to the next one.
    const int sub_row = lane >> 3;
    const int sub_slot = (lane & 7) ^ (((wave & 1) << 2) | (sub_row >> 1));
    const int nn_krow = lane >> 5;
    auto sgpr64 = [](const void* q) {
        const uint64_t u = (uint64_t)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    const uint64_t a_gbase = sgpr64(p.A);
    int i_m0 = 0, i_n0 = 0, i_N = 0, ld_xk = 0, ld_bk = 0;
    uint64_t b_gbase = 0, b_tile_step = 0, xk_base = 0, bk_base = 0;
    unsigned a_off[4], b_off[4];
    auto nn_col = [&](int krow, int ln) {
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        int col = i_n0 + (((ln & 31) ^ (f << 1)) << 3);
        return col + 8 <= i_N ? col : i_N - 8;
    };
    auto set_issue = [&](int m0, int n0, int gi) {
        const uamd_gemm_group& g = p.g[gi];
        i_m0 = m0; i_n0 = n0; i_N = __builtin_amdgcn_readfirstlane(g.N);
        b_gbase = sgpr64(g.B);
        xk_base = sgpr64(g.lora_xk); bk_base = sgpr64(g.lora_bk);
        ld_xk = __builtin_amdgcn_readfirstlane((int)g.ld_xk); ld_bk = __builtin_amdgcn_readfirstlane((int)g.ld_bk);
        const int ldb = __builtin_amdgcn_readfirstlane((int)g.ldb);      // host: ldb < 2^31 / 128
        b_tile_step = BNN ? (uint64_t)ldb * (TK * sizeof(T)) : (uint64_t)(TK * sizeof(T));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int ra = m0 + (c * 8 + wave) * 8 + sub_row;
            ra = ra < M ? ra : M - 1;
            a_off[c] = (unsigned)(((int64_t)ra * p.lda + sub_slot * 8) * (int64_t)sizeof(T));
            if (BNN) {
                // k-row of piece c = c*16 + wave*2 + (lane >> 5): the swizzle term f(krow) does not depend on c, so
                // the pieces differ by the wave-uniform 16 * ldb rows only -- that part rides in the scalar base
                // (issue_main) and ONE per-lane offset serves all four pieces
                const int krow = wave * 2 + nn_krow;
                if (c == 0) b_off[0] = (unsigned)(((int64_t)krow * ldb + nn_col(krow, lane)) * (int64_t)sizeof(T));
            } else {
                int rb = n0 + (c * 8 + wave) * 8 + sub_row;
                rb = rb < i_N ? rb : i_N - 1;
                b_off[c] = (unsigned)(((int64_t)rb * g.ldb + sub_slot * 8) * (int64_t)sizeof(T));
            }
        }
    };

    // ---- LDS addresses of the two stages, swappable (see the header)
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    unsigned dstb[2];
    dstb[0] = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    dstb[1] = dstb[0] + STAGE_BYTES;
    const int frag_off0 = l15 * 128 + ((l4 ^ ((l15 >> 1) & 7)) << 4);
    const int nn_f = (l15 >> 2) | ((l4 & 1) << 2);
    const int nn_lane = 32 * 1024 + (l4 * 8 + (l15 >> 2)) * 512 + (l15 & 3) * 8;
    unsigned fa[2][2], fb[2][4];     // absolute LDS byte addresses (the dynamic region's base included)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[s][ks] = lds_base + s * STAGE_BYTES + (grp * 8) * 2048 + (frag_off0 ^ (ks * 64));
#pragma unroll
        for (int x = 0; x < 4; ++x)
            fb[s][x] = lds_base + (BNN ? s * STAGE_BYTES + nn_lane + (((wn * 4 + x) ^ nn_f) << 5)
                                       : s * STAGE_BYTES + 32 * 1024 + (wn * 4) * 2048 + (frag_off0 ^ ((x & 1) * 64)));
    }
    auto flip_stages = [&]() {
        { const unsigned t = dstb[0]; dstb[0] = dstb[1]; dstb[1] = t; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const unsigned t = fa[0][ks]; fa[0][ks] = fa[1][ks]; fa[1][ks] = t; }
#pragma unroll
        for (int x = 0; x < 4; ++x) { const unsigned t = fb[0][x]; fb[0][x] = fb[1][x]; fb[1][x] = t; }
    };

    auto issue_main = [&](int c, int kt, int stage) {
        const unsigned dst = dstb[stage] + c * 8192;
        if (c < 4) dma16s(a_off[c & 3], a_gbase + (uint64_t)kt * (TK * sizeof(T)), dst);
        else if (BNN) dma16s(b_off[0], b_gbase + (uint64_t)kt * b_tile_step + (uint64_t)(c & 3) * (b_tile_step >> 2), dst);
        else dma16s(b_off[c & 3], b_gbase + (uint64_t)kt * b_tile_step, dst);
    };
    int nk = 0;                      // K tiles of the output tile being multiplied (operands proper + rank block)
    auto issue_any = [&](int c, int kt, int stage) {
        if (kt >= nk) {              // K tile kt - nk of the NEXT output tile: the issue context is already its
            issue_main(c, kt - nk, stage);
        } else if (kt < nk_main) {
            issue_main(c, kt, stage);
        } else {
            const unsigned dst = dstb[stage] + c * 8192;
            // an opaque copy of the lane id: the lane-only parts of these addresses are rebuilt here (a few VALU
            // instructions per output tile) instead of being hoisted to the kernel entry and held -- or spilled --
            // across the whole K loop
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int sub_row = ln >> 3, nn_krow = ln >> 5;
            const int sub_slot = (ln & 7) ^ (((wave & 1) << 2) | (sub_row >> 1));
            if (BNN && c >= 4) {
                const int krow = ((c & 3) * 8 + wave) * 2 + nn_krow;
                const unsigned off = (unsigned)(((int64_t)krow * ld_bk + nn_col(krow, ln)) * (int64_t)sizeof(T));
                dma16s(off, bk_base + (uint64_t)(kt - nk_main) * ((uint64_t)ld_bk * (TK * sizeof(T))), dst);
            } else {
                int row = (c < 4 ? i_m0 : i_n0) + ((c & 3) * 8 + wave) * 8 + sub_row;
                const int last = (c < 4 ? M : i_N) - 1;
                row = row < last ? row : last;
                const int ld = c < 4 ? ld_xk : ld_bk;
                const unsigned off = (unsigned)(((int64_t)row * ld + sub_slot * 8) * (int64_t)sizeof(T));
                dma16s(off, (c < 4 ? xk_base : bk_base) + (uint64_t)(kt - nk_main) * (TK * sizeof(T)), dst);
            }
        }
    };

    frag_t af[4][2], bf[4][2];
    auto read_a = [&](int stage, int mq) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { u32x4_t r; frag_t f; } u;
                u.r = *(const lds_u32x4*)(fa[stage][ks] + (mq * 4 + i) * 2048);
                af[i][ks] = u.f;
            }
    };
    auto read_b = [&](int stage, int nq) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BNN) {
                    union { s16x4_t h[2]; frag_t f; } u;
                    const unsigned a0 = fb[stage][nq * 2 + j] + ks * (32 * 512);
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 512));
                    bf[nq * 2 + j][ks] = u.f;
                } else {
                    union { u32x4_t r; frag_t f; } u;
                    u.r = *(const lds_u32x4*)(fb[stage][ks] + (nq * 2 + j) * 2048);
                    bf[nq * 2 + j][ks] = u.f;
                }
            }
    };
    auto mma = [&](int mq, int nq) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mq * 4 + i][nq * 2 + j] = Mfma2<T>::run(bf[nq * 2 + j][ks], af[i][ks], acc[mq * 4 + i][nq * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto store_tile = [&](int m0, int n0, int gi) {
        const uamd_gemm_group& g = p.g[gi];
        T* Cg = (T*)g.C;
        const int N = g.N;
        const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 7) == 0);
        const T* bias = (const T*)g.bias;
        auto epi = [&](auto acc_c, auto bias_c) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + grp * 128 + i * 16 + l15;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + l4 * 4;
                    if (n >= N) continue;
                    store_c4<T, decltype(acc_c)::value, decltype(bias_c)::value>(
                        Cg + (int64_t)m * g.ldc + n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], n, N, vec_ok, bias);
                }
            }
        };
        if (PLAIN) epi(std::integral_constant<bool, false>{}, std::integral_constant<bool, false>{});
        else UAMD_EPILOGUE_DISPATCH(epi, p.accumulate, bias);
    };

    int v = blockIdx.x;
    int c_m0, c_n0, c_gi;
    decode(v, c_m0, c_n0, c_gi);
    set_issue(c_m0, c_n0, c_gi);
    nk = nk_main + rank_tiles(c_gi);
    // ---- prologue (first output tile of this block only): K tile 0 completely, first 3 pieces of K tile 1
#pragma unroll
    for (int c = 0; c < 8; ++c) issue_main(c, 0, 0);
    issue_main(0, 1, 1); issue_main(1, 1, 1); issue_main(2, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    SLOT_BARRIER();
    if (grp == 1) SLOT_BARRIER();          // anti-phase: group 1 runs one slot behind, for the whole walk

// CHECK = 0: fast loop, every prefetched K tile is a tile of A / B proper of this output tile. CHECK = 1: the last
// K tiles -- the prefetched one may be a rank-block tile, a tile of the next output tile (has_next) or nothing.
#define MORE(CHECK, KT, D) (!(CHECK) || (KT) + (D) < nk || has_next)
#define TILE_P(STAGE, KT, ISSUE, CHECK)                                                   \
    do {                                                                                 \
        /* L0 */                                                                         \
        read_a(STAGE, 0); read_b(STAGE, 0);                                              \
        if (MORE(CHECK, KT, 1)) { ISSUE(3, (KT) + 1, (STAGE) ^ 1); ISSUE(4, (KT) + 1, (STAGE) ^ 1); ISSUE(5, (KT) + 1, (STAGE) ^ 1); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        SLOT_BARRIER();                                                                  \
        mma(0, 0); SLOT_BARRIER();                                                       \
        /* L1 */                                                                         \
        read_b(STAGE, 1);                                                                \
        if (MORE(CHECK, KT, 1)) { ISSUE(6, (KT) + 1, (STAGE) ^ 1); ISSUE(7, (KT) + 1, (STAGE) ^ 1); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        SLOT_BARRIER();                                                                  \
        mma(0, 1); SLOT_BARRIER();                                                       \
        /* L2: the last piece of this output tile has been issued once KT + 2 == nk: hand the issue context over */ \
        read_a(STAGE, 1);                                                                \
        if ((CHECK) && has_next && (KT) + 2 == nk) set_issue(n_m0, n_n0, n_gi);          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        SLOT_BARRIER();                                                                  \
        mma(1, 1); SLOT_BARRIER();                                                       \
        /* L3 */                                                                         \
        if (MORE(CHECK, KT, 2)) {                                                                   \
            ISSUE(0, (KT) + 2, STAGE); ISSUE(1, (KT) + 2, STAGE); ISSUE(2, (KT) + 2, STAGE); \
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                             \
        } else {                                                                         \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
        }                                                                                \
        SLOT_BARRIER();                                                                  \
        mma(1, 0); SLOT_BARRIER();                                                       \
    } while (0)

    for (;;) {
        const int vn = v + (int)gridDim.x;
        const bool has_next = vn < total;
        int n_m0 = 0, n_n0 = 0, n_gi = 0;
        if (has_next) decode(vn, n_m0, n_n0, n_gi);
        int kt = 0;
        for (; kt + 1 < nk_main - 2; kt += 2) {
            TILE_P(0, kt, issue_main, 0);
            TILE_P(1, kt + 1, issue_main, 0);
        }
        for (; kt + 1 < nk; kt += 2) {
            TILE_P(0, kt, issue_any, 1);
            TILE_P(1, kt + 1, issue_any, 1);
        }
        if (kt < nk) TILE_P(0, kt, issue_any, 1);
        store_tile(c_m0, c_n0, c_gi);
        if (!has_next) break;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (nk & 1) flip_stages();
        c_m0 = n_m0; c_n0 = n_n0; c_gi = n_gi;
        nk = nk_main + rank_tiles(c_gi);
        v = vn;
    }
#undef TILE_P
#undef MORE
    if (grp == 0) SLOT_BARRIER();          // match group 1's extra barrier
}

// ------------------------------------------------------------------------------------------------------------
// One wave per SIMD: 4 waves x (128 x 128) wave tiles, the whole accumulator in the 256 AGPRs, and a K loop whose
// instruction ORDER is written out (gemm256s_loop.inc, generated by tools/gen/gen_gemm256s.py -- schedule documented
// there). Why (round 6, profiles/r06_gemm_isa_census.md): per K tile and SIMD the 8-wave ping-pong kernels above issue 349
// instructions for their 128 MFMAs (two waves x {24 ds_read_b128, 8 DMA pieces x 5 instructions, 18 waits, 8 barriers,
// setprio / nop padding}) and read 192 KiB of fragments from LDS; the vendor kernel that is 4-10 % ahead on every NT step
// shape issues 225 and reads 128 KiB -- 128 x 128 wave tiles feed every fragment to eight MFMAs instead of four / eight,
// `buffer_load ... lds` with SGPR piece offsets is one vector + one scalar instruction per DMA piece, and a wave alone on
// its SIMD keeps the matrix pipe's full rate as long as no more than ~3 single-issue instructions sit between two
// MFMAs (profiles/r05_mfma_issue_probe.jsonl) -- which only a hand-ordered stream guarantees. Round 2's 4-wave build
// (hipcc-ordered, full vmcnt(0) drain + one barrier per tile, 3-instruction global_load_lds pieces) measured 4-20 %
// slower than the ping-pong kernel; this one differs in exactly those three points.
// Same LDS tile image, swizzle, XCD raster, rank-block-as-K-tiles and per-accumulator k order as gemm_nt256_kernel, so the
// results are BIT-IDENTICAL to it (tests/test_gpu_nf4_gemm.py). Host contract (gemm256_entry): M % 256 == 0, every
// N_g % 256 == 0 (no edge tiles: the piece offsets ride in SGPRs, rows cannot be clamped per lane), K >= 192.
#include "gemm256s_loop.inc"

template <int N, typename F, int... Is>
__device__ __forceinline__ void g256s_static_for_(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void g256s_static_for(F&& f) {
    g256s_static_for_<N>(f, std::make_integer_sequence<int, N>{});
}

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

// VAR: 0 = the kernel; 2..7 = knock-out timing builds (-DUAMD_G256S_KNOCKOUTS: no DMA / no fragment reads / neither / MFMAs
// only / no vmcnt / no barriers -- results are garbage, tools/gemm_s4_knock.py uses them for nothing but a clock).
// PERSIST: one workgroup per CU walks the virtual block ids v = blockIdx.x, + gridDim.x, ... (same XCD-aware raster) and treats
// the K tiles of consecutive OUTPUT tiles as one stream: the last two K tiles of an output tile are multiplied by the
// steady-state body with its DMA pointed at K tiles 0 and 1 of the NEXT output tile, so a new tile starts with its operands in
// LDS (no cold 64-KiB prologue with every CU bursting at once) and the epilogue's accumulator reads / stores run while that
// data lands. With ONE workgroup per CU (128 KiB of LDS, 512 registers per lane) nothing else can hide a tile's prologue and
// store tail: this is the whole point of the walk.
template <typename T, bool BNN, int VAR, bool PERSIST>
__global__ void __launch_bounds__(256) gemm_nt256s_kernel(G256Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma2<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wave owns rows wm*128.., cols wn*128..
    const int l15 = lane & 15, l4 = lane >> 4;
    const int M = p.M, total = p.total_tiles;

    // virtual block id -> (first row, first column inside its group, group): see gemm_nt256_kernel
    auto decode = [&](int v, int& m0, int& n0, int& gi) {
        int tile;
        {
            const int q = total >> 3, r = total & 7, x = v & 7, j = v >> 3;
            tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        }
        const int gm = p.group_m, tiles_n = p.tile_start[UAMD_G256_MAX_GROUPS];
        const int per_group = gm * tiles_n;
        const int rg = tile / per_group;
        const int first_m = rg * gm;
        const int gsz = min(gm, p.tiles_m - first_m);
        const int rem = tile - rg * per_group;
        const int tn_lin = rem / gsz;
        const int tm = first_m + (rem - tn_lin * gsz);
        int g_ = 0, start = 0;
#pragma unroll
        for (int i = 1; i < UAMD_G256_MAX_GROUPS; ++i)
            if (i < p.n_groups && tn_lin >= p.tile_start[i]) { g_ = i; start = p.tile_start[i]; }
        gi = __builtin_amdgcn_readfirstlane(g_);
        m0 = __builtin_amdgcn_readfirstlane(tm * TM);
        n0 = __builtin_amdgcn_readfirstlane((tn_lin - start) * TN);
    };

    // the accumulators -- quad x * 8 + y = n-tile x, m-tile y of the wave's 128 x 128 -- are a[0:255], owned by the inline asm
    // (gemm256s_loop.inc: named in every MFMA, clobbered by every statement, zeroed / read back by asm helpers;
    // tests/test_attn64_registers.py checks that no compiler-generated instruction touches an AGPR)
    g256s_acc_zero();
    frag_t yf[2][8];                   // [k-half][tile]: A-operand (m) fragments
                                       // (the B-operand's live in the pinned v[192:255], gen_gemm256s.py XREG0)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) yf[h][i] = frag_t{};

    // ---- DMA sources. A-operand (and the NT B-operand): piece c (0..7) issued by wave w fills sub-tile c*4 + w = rows
    //      (c*4 + w)*8 .. +7 of the tile, [8 rows x 64 k] = 8 full lines; lane -> (row = lane >> 3, swizzled 16-byte slot) as
    //      everywhere in this file. The per-lane part (row inside the first piece, slot) is ONE 32-bit offset per operand; the
    //      piece (c * 32 rows) rides in the eight scalar offsets so[c], the K tile in the descriptor's base.
    //      NN B-operand ([K, N], a [64 k][256 n] LDS image): piece = two k-rows x 512 B, sub-tile order chosen so that the bank
    //      swizzle does not depend on c (gen_gemm256s.py PIECE_LDS_NN): k-row of (w, c, h = lane >> 5) = 2 (w & 1) + 8 (w >> 1)
    //      + h + 4 (c & 1) + 16 (c >> 1), f = h | (w & 1) << 1 | (w >> 1) << 2.
    auto u32 = [](int64_t v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v); };
    auto sgpr64 = [](const void* q, int64_t byte_off) {      // a wave-uniform address as a value the compiler keeps in SGPRs
        const uint64_t u = (uint64_t)(uintptr_t)q + (uint64_t)byte_off;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    // bases of the two buffer descriptors (tile origin + the K tiles fetched so far), as 32-bit halves
    unsigned curAlo = 0, curAhi = 0, curBlo = 0, curBhi = 0;
    unsigned voffA = 0, voffB = 0, soA[8], soB[8], stepA = TK * sizeof(T), stepB = TK * sizeof(T);
    // the lane-only parts are rebuilt at every switch from an opaque copy of the lane id (a few VALU instructions per output
    // tile) instead of being held -- or spilled -- across the K loop
    // `ahead`: the descriptors are pointed at K tile `ahead` of these matrices (2 at the top of an output tile: its K tiles 0 and
    // 1 were fetched by the prologue / during the previous tile's last two K tiles -- recomputed, not carried over: a value that
    // is written by asm AND lives across the tile loop's back edge is a PHI hipcc handles through VGPRs, and the first build
    // that carried the bases that way read garbage for every second output tile)
    auto set_sources = [&](const void* Ap, int ld_a, const void* Bp, int ld_b, int m0, int n0, int ahead) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int sr = ln >> 3, ss = (ln & 7) ^ (((wave & 1) << 2) | (sr >> 1));
        uint64_t baseA = (uint64_t)(uintptr_t)Ap + (uint64_t)((int64_t)m0 * ld_a * (int64_t)sizeof(T)) +
                         (uint64_t)ahead * (TK * sizeof(T));
        uint64_t baseB;
        voffA = (unsigned)(((wave * 8 + sr) * ld_a + ss * 8) * (int)sizeof(T));
        if constexpr (BNN) {
            const int h = ln >> 5, f = h | ((wave & 1) << 1) | ((wave >> 1) << 2);
            const int krow0 = 2 * (wave & 1) + 8 * (wave >> 1) + h;
            baseB = (uint64_t)(uintptr_t)Bp + (uint64_t)((int64_t)n0 * (int64_t)sizeof(T));
            voffB = (unsigned)((krow0 * ld_b + (((ln & 31) ^ (f << 1)) << 3)) * (int)sizeof(T));
            stepB = u32((int64_t)TK * ld_b * (int64_t)sizeof(T));
        } else {
            baseB = (uint64_t)(uintptr_t)Bp + (uint64_t)((int64_t)n0 * ld_b * (int64_t)sizeof(T));
            voffB = (unsigned)(((wave * 8 + sr) * ld_b + ss * 8) * (int)sizeof(T));
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            soA[c] = u32((int64_t)c * 32 * ld_a * (int64_t)sizeof(T));
            soB[c] = BNN ? u32((int64_t)(4 * (c & 1) + 16 * (c >> 1)) * ld_b * (int64_t)sizeof(T)) : u32((int64_t)c * 32 * ld_b * (int64_t)sizeof(T));
        }
        baseB += (uint64_t)ahead * stepB;
        {
            curAlo = __builtin_amdgcn_readfirstlane((unsigned)baseA);
            curAhi = __builtin_amdgcn_readfirstlane((unsigned)(baseA >> 32)) & 0xffffu;     // (descriptor word 1: stride 0)
            curBlo = __builtin_amdgcn_readfirstlane((unsigned)baseB);
            curBhi = __builtin_amdgcn_readfirstlane((unsigned)(baseB >> 32)) & 0xffffu;
        }
    };
    auto set_main = [&](int m0, int n0, int gi, int ahead) {
        const uamd_gemm_group& g = p.g[gi];
        set_sources(p.A, __builtin_amdgcn_readfirstlane((int)p.lda), g.B, __builtin_amdgcn_readfirstlane((int)g.ldb), m0, n0, ahead);
    };
    auto set_rank = [&](int m0, int n0, int gi) {
        const uamd_gemm_group& g = p.g[gi];
        set_sources(g.lora_xk, __builtin_amdgcn_readfirstlane((int)g.ld_xk), g.lora_bk, __builtin_amdgcn_readfirstlane((int)g.ld_bk), m0, n0, 0);
    };
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;      // 0: no static LDS in this kernel (stage bit = 0x10000)
    // DMA destinations of the wave's first piece of each operand in stage `par`, + 64 (gen_gemm256s.py M0_BIAS). Like the
    // descriptor bases they are rebuilt at the top of every output tile (K tile 2 goes where K tile 0 sits), not carried over
    unsigned m0bA = 0, m0bB = 0;
    auto make_m0 = [&](int par) {
        const unsigned st = lds_base + ((unsigned)par << 16) + 64;
        m0bA = __builtin_amdgcn_readfirstlane(st + wave * 1024);
        m0bB = __builtin_amdgcn_readfirstlane(st + 32 * 1024 + (BNN ? (wave & 1) * 1024 + (wave >> 1) * 4096 : wave * 1024));
    };
    make_m0(0);
    // fragment read pointers of the stage that holds K tile 0 of the current output tile (`par`), rebuilt per output tile from an
    // opaque copy of the lane id: held across the epilogue (which wants every VGPR for the accumulators) they were spilled
    unsigned rdA[2], rdB[2], rdBn[8];
    auto make_rd = [&](int par) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int a15 = ln & 15, a4 = ln >> 4;
        const unsigned st = lds_base + ((unsigned)par << 16);
        const int frag_off0 = a15 * 128 + ((a4 ^ ((a15 >> 1) & 7)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            rdA[h] = st + (wm * 8) * 2048 + (frag_off0 ^ (h * 64));
            rdB[h] = st + 32 * 1024 + (wn * 8) * 2048 + (frag_off0 ^ (h * 64));
        }
        // transposing reads: k-row l4*8 + (l15 >> 2) (+4 for the second read, + 32 for the second k-half), logical 32-byte
        // granule (16-column tile) wn*8 + x, swizzled by the k-row's f
        const int nn_f = (a15 >> 2) | ((a4 & 1) << 2);
        const int nn_lane = 32 * 1024 + (a4 * 8 + (a15 >> 2)) * 512 + (a15 & 3) * 8;
#pragma unroll
        for (int x = 0; x < 8; ++x) rdBn[x] = st + nn_lane + (((wn * 8 + x) ^ nn_f) << 5);
    };
    const int nk_main = __builtin_amdgcn_readfirstlane(p.K / TK);       // host: >= 3
    unsigned cnt = 0;

#define G256S_ASM(BODY)                                                                              \
    asm volatile(BODY : G256S_OUT_FRAGS, G256S_OUT_RD, G256S_OUT_M0, [cnt] "+s"(cnt)   \
                 : G256S_IN_SO, G256S_IN_DMA : G256S_CLOBBER)
#define G256SN_ASM(BODY)                                                                             \
    asm volatile(BODY : G256SN_OUT_FRAGS, G256SN_OUT_RD, G256S_OUT_M0, [cnt] "+s"(cnt) \
                 : G256S_IN_SO, G256S_IN_DMA : G256SN_CLOBBER)
// bodies without DMA take no DMA operands: the values the rank-block branch below rewrites are then dead at its join
// (as live-out "s" operands they become PHIs, which hipcc refuses to keep in SGPRs: "illegal VGPR to SGPR copy")
#define G256S_ASM_C(BODY) asm volatile(BODY : G256S_OUT_FRAGS, G256S_OUT_RD : : G256S_CLOBBER_C)
#define G256SN_ASM_C(BODY) asm volatile(BODY : G256SN_OUT_FRAGS, G256SN_OUT_RD : : G256SN_CLOBBER_C)
// the first fragment reads of an output tile: every fragment is WRITE-ONLY here, so nothing of the previous tile's stays live
// across the epilogue in front of it
#define G256S_ASM_W(BODY) asm volatile(BODY : G256S_OUTW_FRAGS, G256S_OUT_RD : : G256S_CLOBBER_C)
#define G256SN_ASM_W(BODY) asm volatile(BODY : G256SN_OUTW_FRAGS, G256SN_OUT_RD : : G256SN_CLOBBER_C)
// NAME: the part of the macro name behind G256S_ / G256SN_; K: ASM (DMA operands), ASM_C or ASM_W
#define G256S_RUN2(NAME, K)                                                                 \
    do {                                                                                    \
        if constexpr (BNN) {                                                                \
            if constexpr (std::is_same<T, bf16_t>::value) G256SN_##K(G256SN_##NAME("bf16")); \
            else G256SN_##K(G256SN_##NAME("f16"));                                          \
        } else {                                                                            \
            if constexpr (std::is_same<T, bf16_t>::value) G256S_##K(G256S_##NAME("bf16"));  \
            else G256S_##K(G256S_##NAME("f16"));                                            \
        }                                                                                   \
    } while (0)
// (the descriptor bases are assigned under wave-uniform branches AND written by asm: hipcc treats an asm output as divergent, the
// PHI at the join becomes a VGPR -- readfirstlane hands the next statement a scalar again)
#define G256S_RUN(NAME)                                                                                     \
    do {                                                                                                    \
        curAlo = __builtin_amdgcn_readfirstlane(curAlo); curAhi = __builtin_amdgcn_readfirstlane(curAhi);   \
        curBlo = __builtin_amdgcn_readfirstlane(curBlo); curBhi = __builtin_amdgcn_readfirstlane(curBhi);   \
        G256S_RUN2(NAME, ASM);                                                                              \
    } while (0)
#define G256S_RUN_C(NAME) G256S_RUN2(NAME, ASM_C)
#define G256S_RUN_W(NAME) G256S_RUN2(NAME, ASM_W)

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+l15, n = x*16 + 4*l4 (B-operand is MFMA source A) for each of its 64
    //      accumulator quads. Stored as 8-byte pieces that is 64 store instructions per lane, each touching 16 rows with 32 bytes
    //      -- and with one workgroup per CU the tile's store tail hides behind nothing (round 3's knock-out: half the store
    //      instructions = +7.5 % on o_proj). The lane pairs (l4, l4 ^ 1) -- lanes l and l ^ 16, the 16-lane ROWS that
    //      v_permlane16_swap exchanges -- hold adjacent column quads of the SAME row: one swap per packed dword of a tile pair
    //      (x even, x + 1) hands the even row both quads of tile x and the odd row both of tile x + 1: 32 stores of 16 bytes,
    //      64 contiguous bytes per row and instruction, same bytes at the same addresses (host contract: whole tiles, so no
    //      edge tests; ldc % 8 == 0 and a 16-byte aligned C checked there too).
    auto store_tile = [&](int m0, int n0, int gi) {
        const uamd_gemm_group& g = p.g[gi];
        T* Cg = (T*)g.C;
        const T* bias = (const T*)g.bias;
        int lno = lane;
        asm volatile("" : "+v"(lno));            // (rebuilt per tile, see set_sources)
        const int l4o = lno >> 4, l15o = lno & 15, odd = l4o & 1;
        auto epi = [&](auto acc_c, auto bias_c) {
            constexpr bool ACC = decltype(acc_c)::value, BIAS = decltype(bias_c)::value;
            g256s_static_for<32>([&](auto idx) {
                constexpr int y = decltype(idx)::value >> 2, xp = decltype(idx)::value & 3;
                const int m = m0 + wm * 128 + y * 16 + l15o;
                T* crow = Cg + (int64_t)m * g.ldc + n0 + wn * 128;
                float f[8];                                 // [tile of the pair][column of the quad]
                g256s_acc_read<y, xp>(f);
                uint32_t w[2][2];                           // [tile of the pair][dword]: this lane's quads, packed
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int x = 2 * xp + t;
                    const int nq = x * 16 + l4o * 4;        // column of the quad inside the wave's 128
                    if (BIAS) {
                        union { uint2 raw; T e[4]; } bv;
                        bv.raw = *reinterpret_cast<const uint2*>(bias + n0 + wn * 128 + nq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[4 * t + r] += to_f32(bv.e[r]);
                    }
                    if (ACC) {
                        union { uint2 raw; T e[4]; } cv;
                        cv.raw = *reinterpret_cast<const uint2*>(crow + nq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[4 * t + r] += to_f32(cv.e[r]);
                    }
                    union { T e[4]; uint32_t u[2]; } o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.e[r] = from_f32<T>(f[4 * t + r]);
                    w[t][0] = o.u[0];
                    w[t][1] = o.u[1];
                }
                // odd rows of the first operand <-> even rows of the second: even lanes keep their tile-x quad and receive
                // the partner's, odd lanes keep their tile-(x+1) quad and receive the partner's
                const auto s0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
                uint4 out;
                out.x = s0[0]; out.y = s1[0]; out.z = s0[1]; out.w = s1[1];
                // even lane: [own tile-x quad | partner's tile-x quad] at tile x, column 4 l4; odd lane: [partner's
                // tile-(x+1) quad | own] at tile x + 1, column 4 (l4 - 1)
                *reinterpret_cast<uint4*>(crow + (2 * xp + odd) * 16 + (l4o & 2) * 4) = out;
            });
        };
        UAMD_EPILOGUE_DISPATCH(epi, p.accumulate, bias);
    };

    int v = blockIdx.x;
    int c_m0, c_n0, c_gi;
    int par = 0;                       // LDS stage of the current output tile's K tile 0
    decode(v, c_m0, c_n0, c_gi);
    set_main(c_m0, c_n0, c_gi, 0);
    make_rd(0);
    // ---- prologue (first output tile of this workgroup only): K tiles 0 and 1 in flight, tile 0 landed and published
    G256S_RUN(ISSUE_TILE);
    G256S_RUN(ISSUE_TILE);
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    for (;;) {
        // (the per-lane source offsets and read pointers of THIS tile, rebuilt here: nothing per-lane lives across the epilogue)
        set_main(c_m0, c_n0, c_gi, 2);
        make_rd(par);
        make_m0(par);
        G256S_RUN_W(READ0);            // k-half-0 fragments of this output tile's K tile 0
        // ---- tiles 0 .. nk_main - 3 fetch tiles 2 .. nk_main - 1 of the operands proper (the loop runs cnt + 1 trips)
        //      -- unconditional (host: K >= 192): a branch around an asm statement with "+s" operands makes them PHIs
        cnt = (unsigned)(nk_main - 3);
        if constexpr (BNN || VAR == 0) {
            G256S_RUN(LOOP);
        } else {
            if constexpr (VAR == 2) G256S_ASM(G256S_LOOP_KND("bf16"));
            else if constexpr (VAR == 3) G256S_ASM(G256S_LOOP_KNR("bf16"));
            else if constexpr (VAR == 4) G256S_ASM(G256S_LOOP_KMF("bf16"));
            else if constexpr (VAR == 5) G256S_ASM(G256S_LOOP_KMO("bf16"));
            else if constexpr (VAR == 6) G256S_ASM(G256S_LOOP_KNV("bf16"));
            else G256S_ASM(G256S_LOOP_KNB("bf16"));
        }
        // ---- the rank block's tiles are fetched from XK [M, Rk] / BK ([N, Rk]; NN: [Rk, N]) by the same body: only the sources change
        const int nk_rank = __builtin_amdgcn_readfirstlane(p.g[c_gi].lora_xk != nullptr ? p.g[c_gi].Rk / TK : 0);
        if (nk_rank) {
            set_rank(c_m0, c_n0, c_gi);
            cnt = (unsigned)(nk_rank - 1);
            G256S_RUN(LOOP);
        }
        const int vn = PERSIST ? v + (int)gridDim.x : total;
        const bool has_next = PERSIST && vn < total;
        int n_m0 = 0, n_n0 = 0, n_gi = 0;
        if (has_next) {
            // ---- the last two K tiles of this output tile, multiplied while K tiles 0 and 1 of the NEXT one are fetched
            decode(vn, n_m0, n_n0, n_gi);
            set_main(n_m0, n_n0, n_gi, 0);
            cnt = 1;
            G256S_RUN(LOOP);
        } else {
            G256S_RUN_C(NODMA);
            G256S_RUN_C(LAST);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // last MFMA's write -> the epilogue's v_accvgpr_read (asm MFMAs)
        store_tile(c_m0, c_n0, c_gi);
        if (!has_next) break;
        g256s_acc_zero();
        par ^= (nk_main + nk_rank) & 1;       // every K tile flipped the stages once
        c_m0 = n_m0; c_n0 = n_n0; c_gi = n_gi;
        v = vn;
    }
#undef G256S_RUN_W
#undef G256S_RUN_C
#undef G256S_RUN
#undef G256S_RUN2
#undef G256SN_ASM_W
#undef G256S_ASM_W
#undef G256SN_ASM_C
#undef G256S_ASM_C
#undef G256SN_ASM
#undef G256S_ASM
}

// ------------------------------------------------------------------------------------------------------------
// Half-height variant: 128 x 256 x 64 tiles, same LDS-DMA ping-pong, for launches whose 256 x 256 tiling would leave
// CUs idle (M = 2048 tokens: o_proj / down_proj / every dX GEMM have 8 x 16 = 128 such tiles for 256 CUs; the
// 128-row tiling gives 256). Each wave group owns 64 rows (wave tile 64 x 64, 32 MFMAs per K tile), a K tile is
// four slots (L0 M0 L1 M1) and the LDS ring has THREE 48 KiB stages: the six DMA pieces of tile t+2 are issued
// during tile t (3 in L0, 3 in L1, into the stage tile t-1 was read from: both groups finished it one barrier
// ago), and tile t+1 is waited for with a counted vmcnt(6) at the end of L1 -- one slot before the tile ends,
// so the group that runs one slot behind has passed its own wait before the other one starts reading.
constexpr int TMH = 128;
constexpr int STAGE_H = (TMH + TN) * TK * 2;       // 48 KiB
constexpr int LDS_H = 3 * STAGE_H;                 // 144 KiB

template <typename T, bool BNN>
__global__ void __launch_bounds__(512, 2) gemm_nt256h_kernel(G256Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma2<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;      // group owns rows grp*64.., wave owns cols wn*64..
    const int l15 = lane & 15, l4 = lane >> 4;
    int tile = blockIdx.x;
    {
        const int nt = p.total_tiles, q = nt >> 3, r = nt & 7, x = tile & 7, j = tile >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    int tm, tn_lin;
    {
        const int gm = p.group_m, tiles_n = p.tile_start[UAMD_G256_MAX_GROUPS];
        const int per_group = gm * tiles_n;
        const int rg = tile / per_group;
        const int first_m = rg * gm;
        const int gsz = min(gm, p.tiles_m - first_m);
        const int rem = tile - rg * per_group;
        tn_lin = rem / gsz;
        tm = first_m + (rem - tn_lin * gsz);
    }
    int gi = 0;
#pragma unroll
    for (int i = 1; i < UAMD_G256_MAX_GROUPS; ++i)
        if (i < p.n_groups && tn_lin >= p.tile_start[i]) gi = i;
    const uamd_gemm_group& g = p.g[gi];
    const int m0 = tm * TMH, n0 = (tn_lin - p.tile_start[gi]) * TN;
    const int M = p.M, K = p.K, N = g.N;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // piece c (0..5) of a tile, issued by wave w, fills sub-tile u = c*8 + w: u < 16 -> A rows u*8.., else B rows (u-16)*8..
    const int sub_row = lane >> 3;
    const int sub_slot = (lane & 7) ^ (((wave & 1) << 2) | (sub_row >> 1));
    auto sgpr64 = [](const void* q) {
        const uint64_t u = (uint64_t)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    const uint64_t a_gbase = sgpr64(p.A), b_gbase = sgpr64(g.B);
    unsigned a_off[2], b_off[4];
    const int nn_krow = lane >> 5;                  // BNN: see gemm_nt256_kernel
    auto nn_col = [&](int krow) {
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        int col = n0 + (((lane & 31) ^ (f << 1)) << 3);
        return col + 8 <= N ? col : N - 8;
    };
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int ra = m0 + (c * 8 + wave) * 8 + sub_row;
        ra = ra < M ? ra : M - 1;
        a_off[c] = (unsigned)(((int64_t)ra * p.lda + sub_slot * 8) * (int64_t)sizeof(T));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (BNN) {
            const int krow = (c * 8 + wave) * 2 + nn_krow;
            b_off[c] = (unsigned)(((int64_t)krow * g.ldb + nn_col(krow)) * (int64_t)sizeof(T));
        } else {
            int rb = n0 + (c * 8 + wave) * 8 + sub_row;
            rb = rb < N ? rb : N - 1;
            b_off[c] = (unsigned)(((int64_t)rb * g.ldb + sub_slot * 8) * (int64_t)sizeof(T));
        }
    }
    const uint64_t b_tile_step = BNN ? (uint64_t)__builtin_amdgcn_readfirstlane((int)g.ldb) * (TK * sizeof(T))
                                     : (uint64_t)(TK * sizeof(T));
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const int nk_main = __builtin_amdgcn_readfirstlane(K / TK);
    const int nk = __builtin_amdgcn_readfirstlane(nk_main + (g.lora_xk != nullptr ? g.Rk / TK : 0));
    auto issue_main = [&](int c, int kt, int stage) {       // c, stage compile-time
        const unsigned dst = lds_base + stage * STAGE_H + (c * 8 + wave) * 1024;
        if (c < 2) dma16s(a_off[c], a_gbase + (uint64_t)kt * (TK * sizeof(T)), dst);
        else dma16s(b_off[c - 2], b_gbase + (uint64_t)kt * b_tile_step, dst);
    };
    const uint64_t xk_base = sgpr64(g.lora_xk), bk_base = sgpr64(g.lora_bk);
    const int ld_xk = __builtin_amdgcn_readfirstlane((int)g.ld_xk), ld_bk = __builtin_amdgcn_readfirstlane((int)g.ld_bk);
    auto issue_any = [&](int c, int kt, int stage) {
        if (kt < nk_main) {
            issue_main(c, kt, stage);
        } else {
            const unsigned dst = lds_base + stage * STAGE_H + (c * 8 + wave) * 1024;
            if (BNN && c >= 2) {
                const int krow = ((c - 2) * 8 + wave) * 2 + nn_krow;
                const unsigned off = (unsigned)(((int64_t)krow * ld_bk + nn_col(krow)) * (int64_t)sizeof(T));
                dma16s(off, bk_base + (uint64_t)(kt - nk_main) * ((uint64_t)ld_bk * (TK * sizeof(T))), dst);
            } else {
                int row = (c < 2 ? m0 + (c * 8 + wave) * 8 : n0 + ((c - 2) * 8 + wave) * 8) + sub_row;
                const int last = (c < 2 ? M : N) - 1;
                row = row < last ? row : last;
                const int ld = c < 2 ? ld_xk : ld_bk;
                const unsigned off = (unsigned)(((int64_t)row * ld + sub_slot * 8) * (int64_t)sizeof(T));
                dma16s(off, (c < 2 ? xk_base : bk_base) + (uint64_t)(kt - nk_main) * (TK * sizeof(T)), dst);
            }
        }
    };
    const int frag_off0 = l15 * 128 + ((l4 ^ ((l15 >> 1) & 7)) << 4);
    const int frag_off[2] = {frag_off0, frag_off0 ^ 64};
    const int a_lbase = (grp * 4) * 2048;
    const int b_lbase = 16 * 1024 + (wn * 4) * 2048;
    frag_t af[4][2], bf[4][2];
    auto read_a = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { uint4 r; frag_t f; } u;
                u.r = *reinterpret_cast<const uint4*>(smem + stage * STAGE_H + a_lbase + i * 2048 + frag_off[ks]);
                af[i][ks] = u.f;
            }
    };
    const int nn_f = (l15 >> 2) | ((l4 & 1) << 2);
    const int nn_lane = 16 * 1024 + (l4 * 8 + (l15 >> 2)) * 512 + (l15 & 3) * 8;
    auto read_b = [&](int stage, int nq) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BNN) {
                    union { s16x4_t h[2]; frag_t f; } u;
                    const int a0 = stage * STAGE_H + nn_lane + ks * (32 * 512) + (((wn * 4 + nq * 2 + j) ^ nn_f) << 5);
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + a0));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + a0 + 4 * 512));
                    bf[nq * 2 + j][ks] = u.f;
                } else {
                    union { uint4 r; frag_t f; } u;
                    u.r = *reinterpret_cast<const uint4*>(smem + stage * STAGE_H + b_lbase + (nq * 2 + j) * 2048 + frag_off[ks]);
                    bf[nq * 2 + j][ks] = u.f;
                }
            }
    };
    auto mma = [&](int nq) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][nq * 2 + j] = Mfma2<T>::run(bf[nq * 2 + j][ks], af[i][ks], acc[i][nq * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    // ---- prologue: tiles 0 and 1 completely
#pragma unroll
    for (int c = 0; c < 6; ++c) issue_main(c, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int c = 0; c < 6; ++c) issue_any(c, 1, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SLOT_BARRIER();
    if (grp == 1) SLOT_BARRIER();          // anti-phase: group 1 runs one slot behind

#define NEXT2(ST) ((ST) == 0 ? 2 : (ST) - 1)     /* stage of tile kt + 2 = (ST + 2) % 3 */
#define TILE_H(STAGE, KT, ISSUE, CHECK)                                                   \
    do {                                                                                 \
        /* L0 */                                                                         \
        read_a(STAGE); read_b(STAGE, 0);                                                 \
        if (!(CHECK) || (KT) + 2 < nk) { ISSUE(0, (KT) + 2, NEXT2(STAGE)); ISSUE(1, (KT) + 2, NEXT2(STAGE)); ISSUE(2, (KT) + 2, NEXT2(STAGE)); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        SLOT_BARRIER();                                                                  \
        mma(0); SLOT_BARRIER();                                                          \
        /* L1 */                                                                         \
        read_b(STAGE, 1);                                                                \
        if (!(CHECK) || (KT) + 2 < nk) {                                                 \
            ISSUE(3, (KT) + 2, NEXT2(STAGE)); ISSUE(4, (KT) + 2, NEXT2(STAGE)); ISSUE(5, (KT) + 2, NEXT2(STAGE)); \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(6)" ::: "memory");     \
        } else {                                                                         \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");     \
        }                                                                                \
        SLOT_BARRIER();                                                                  \
        mma(1); SLOT_BARRIER();                                                          \
    } while (0)

    int kt = 0;
    // fast loop: tiles kt <= nk_main - 3 prefetch tile kt + 2 <= nk_main - 1 (a tile of A / B proper)
    for (; kt + 2 < nk_main - 2; kt += 3) {
        TILE_H(0, kt, issue_main, 0);
        TILE_H(1, kt + 1, issue_main, 0);
        TILE_H(2, kt + 2, issue_main, 0);
    }
    for (; kt < nk; kt += 3) {             // kt is a multiple of 3 here: stage = kt % 3 stays compile-time
        TILE_H(0, kt, issue_any, 1);
        if (kt + 1 < nk) TILE_H(1, kt + 1, issue_any, 1);
        if (kt + 2 < nk) TILE_H(2, kt + 2, issue_any, 1);
    }
#undef TILE_H
#undef NEXT2
    if (grp == 0) SLOT_BARRIER();          // match group 1's extra barrier

    T* Cg = (T*)g.C;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 7) == 0);
    const T* bias = (const T*)g.bias;
    auto epi = [&](auto acc_c, auto bias_c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + grp * 64 + i * 16 + l15;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + l4 * 4;
                if (n >= N) continue;
                store_c4<T, decltype(acc_c)::value, decltype(bias_c)::value>(
                    Cg + (int64_t)m * g.ldc + n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], n, N, vec_ok, bias);
            }
        }
    };
    UAMD_EPILOGUE_DISPATCH(epi, p.accumulate, bias);
}

template <typename T, bool BNN>
int launch256h(const G256Args& a, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256h_kernel<T, BNN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_H);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_nt256h_kernel<T, BNN>), dim3((unsigned)a.total_tiles), dim3(512), LDS_H, st, a);
    return uamd_launch_status();
}

template <typename T, bool BNN, int VAR, bool PERSIST>
int launch256s_(const G256Args& a, hipStream_t st, int grid) {
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256s_kernel<T, BNN, VAR, PERSIST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_nt256s_kernel<T, BNN, VAR, PERSIST>), dim3((unsigned)grid), dim3(256), LDS_BYTES, st, a);
    return uamd_launch_status();
}

int cu_count();

// UAMD_TUNE_GEMM_S: 1 = by tile count (the persistent walk when every CU gets at least FOUR output tiles, one workgroup per tile
// otherwise), 2 = one workgroup per tile always, 9 = the walk from two tiles per CU on (tests), 0 = the 8-wave kernels
// (gemm256_entry); 3..8 = knock-out builds. Measured (profiles/r06l_gemm_persistent_ab.jsonl, same box, interleaved with
// hipBLASLt): gate|up (14 tiles per CU) +1.3 %, down-dX (7) +1.1 %, 2-3 tiles per CU -1.1 .. 0 % -- the dispatcher's balancing is
// worth more than the hidden prologue there.
template <typename T, bool BNN>
int launch256s(const G256Args& a, hipStream_t st) {
    const int v = uamd_tuning_get(UAMD_TUNE_GEMM_S);
#ifdef UAMD_G256S_KNOCKOUTS
    if constexpr (std::is_same<T, bf16_t>::value && !BNN) {
        if (v == 3) return launch256s_<T, false, 2, false>(a, st, a.total_tiles);
        if (v == 4) return launch256s_<T, false, 3, false>(a, st, a.total_tiles);
        if (v == 5) return launch256s_<T, false, 4, false>(a, st, a.total_tiles);
        if (v == 6) return launch256s_<T, false, 5, false>(a, st, a.total_tiles);
        if (v == 7) return launch256s_<T, false, 6, false>(a, st, a.total_tiles);
        if (v == 8) return launch256s_<T, false, 7, false>(a, st, a.total_tiles);
    }
#endif
    const int n_cu = cu_count();
    if (v != 2 && (n_cu & 7) == 0 && a.total_tiles >= (v == 9 ? 2 : 4) * n_cu) return launch256s_<T, BNN, 0, true>(a, st, n_cu);
    return launch256s_<T, BNN, 0, false>(a, st, a.total_tiles);
}

template <typename T, bool BNN, bool ATN = false>
int launch256(const G256Args& a, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256_kernel<T, BNN, ATN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_nt256_kernel<T, BNN, ATN>), dim3((unsigned)a.total_tiles), dim3(512), LDS_BYTES, st, a);
    return uamd_launch_status();
}

template <typename T, bool BNN, bool PLAIN>
int launch256p_(const G256Args& a, hipStream_t st, int n_cu) {
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt256p_kernel<T, BNN, PLAIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    const int grid = a.total_tiles < n_cu ? a.total_tiles : n_cu;
    hipLaunchKernelGGL((gemm_nt256p_kernel<T, BNN, PLAIN>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st, a);
    return uamd_launch_status();
}

template <typename T, bool BNN>
int launch256p(const G256Args& a, hipStream_t st, int n_cu) {
    bool plain = !a.accumulate && uamd_tuning_get(UAMD_TUNE_GEMM_PLAIN) != 0;
    for (int i = 0; i < a.n_groups; ++i) plain = plain && a.g[i].bias == nullptr;
    return plain ? launch256p_<T, BNN, true>(a, st, n_cu) : launch256p_<T, BNN, false>(a, st, n_cu);
}

// compute units of the current device (one persistent block each: 128 KiB of the CU's 160 KiB LDS)
int cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}

}  // namespace

#ifdef UAMD_G256_TRACE
extern "C" int uamd_debug_g256_trace(unsigned* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace256), &buf, sizeof(buf));
}
#endif

// Same contract as uamd_gemm_nt (dense B), 256x256x64 tiles. Requires K % 64 == 0. The LoRA term comes as the
// rank block lora_xk / lora_bk (extra K tiles); a group that only carries lora_xa / lora_b is rejected.
static int gemm256_entry(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups, int n_groups,
                         int accumulate, int dtype, void* stream, bool bnn, bool atn = false) {
    if (M < 0 || K <= 0 || n_groups < 1 || n_groups > UAMD_G256_MAX_GROUPS || !groups) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 63) || (lda & 7) || !aligned16(A)) return UAMD_ERR_ALIGN;
    // per-lane addresses are 32-bit byte offsets from the matrix base: every operand must span < 4 GiB
    const int64_t kSpan = (int64_t)1 << 32;
    if (atn) {                                  // A [K, M]: whole 16-byte slots of M, K rows x lda below 4 GiB
        if (!bnn) return UAMD_ERR_ARG;
        if ((M & 7) || M < 8) return UAMD_ERR_ALIGN;
        if ((int64_t)K * lda * 2 >= kSpan || lda > 0x7fffffffLL / 128) return UAMD_ERR_ARG;
    } else if ((int64_t)M * lda * 2 >= kSpan) {
        return UAMD_ERR_ARG;
    }
    G256Args a;
    a.A = A; a.lda = lda; a.M = M; a.K = K; a.n_groups = n_groups; a.accumulate = accumulate;
    // tile height: 256 rows when that tiling fills the chip, else 128 rows (twice the tiles; UAMD_TUNE_GEMM_HALF:
    // 0 = never, 1 = when the 256-row tiling has fewer than 192 tiles (default), 2 = always)
    int tn_all = 0;
    for (int i = 0; i < n_groups; ++i) tn_all += (groups[i].N + TN - 1) / TN;
    const int half_mode = uamd_tuning_get(UAMD_TUNE_GEMM_HALF);
    const bool half = !atn && (half_mode == 2 || (half_mode == 1 && (int64_t)((M + TM - 1) / TM) * tn_all < 192));
    const int tile_m = half ? TMH : TM;
    a.tiles_m = (M + tile_m - 1) / tile_m;
    int tn = 0;
    for (int i = 0; i < UAMD_G256_MAX_GROUPS; ++i) {
        a.tile_start[i] = tn;
        if (i < n_groups) {
            const uamd_gemm_group& g = groups[i];
            if (g.N <= 0 || !g.B || !g.C) return UAMD_ERR_ARG;
            if ((g.ldb & 7) || !aligned16(g.B)) return UAMD_ERR_ALIGN;
            if (bnn) {                          // B_g [K, N]: N % 8 == 0 (whole 16-byte slots), rows x ldb below 4 GiB
                if ((g.N & 7) || g.N < 8) return UAMD_ERR_ALIGN;
                if ((int64_t)K * g.ldb * 2 >= kSpan || g.ldb > 0x7fffffffLL / 128) return UAMD_ERR_ARG;
            } else if ((int64_t)g.N * g.ldb * 2 >= kSpan) {
                return UAMD_ERR_ARG;
            }
            if (g.lora_xa && !g.lora_xk) return UAMD_ERR_ARG;     // this kernel takes the rank block as K tiles
            if (atn && (g.lora_xk || g.lora_xa)) return UAMD_ERR_ARG;
            if (g.lora_xk) {
                if (!g.lora_bk || g.Rk <= 0) return UAMD_ERR_ARG;
                if ((g.Rk & 63) || (g.ld_xk & 7) || (g.ld_bk & 7) || !aligned16(g.lora_xk) || !aligned16(g.lora_bk))
                    return UAMD_ERR_ALIGN;
                if ((int64_t)M * g.ld_xk * 2 >= kSpan) return UAMD_ERR_ARG;
                if ((int64_t)(bnn ? g.Rk : g.N) * g.ld_bk * 2 >= kSpan || g.ld_bk > 0x7fffffffLL / 128) return UAMD_ERR_ARG;
            }
            a.g[i] = g;
            tn += (g.N + TN - 1) / TN;
        } else {
            a.g[i] = groups[0];
        }
    }
    a.tile_start[UAMD_G256_MAX_GROUPS] = tn;
    const int64_t total = (int64_t)tn * a.tiles_m;
    if (total > 0x7fffffffLL) return UAMD_ERR_ARG;
    a.total_tiles = (int)total;
    {
        const int gm = uamd_tuning_get(UAMD_TUNE_GROUP_M);
        a.group_m = gm < 1 ? 1 : (gm < a.tiles_m ? gm : a.tiles_m);
    }
    hipStream_t st = (hipStream_t)stream;
    if (atn) {
        if (dtype == UAMD_BF16) return launch256<bf16_t, true, true>(a, st);
        if (dtype == UAMD_F16) return launch256<f16_t, true, true>(a, st);
        return UAMD_ERR_DTYPE;
    }
    if (half) {
        if (dtype == UAMD_BF16) return bnn ? launch256h<bf16_t, true>(a, st) : launch256h<bf16_t, false>(a, st);
        if (dtype == UAMD_F16) return bnn ? launch256h<f16_t, true>(a, st) : launch256h<f16_t, false>(a, st);
        return UAMD_ERR_DTYPE;
    }
    // whole-tile NT / NN launches: the one-wave-per-SIMD kernel (UAMD_TUNE_GEMM_S)
    if (K >= 3 * TK && (M & (TM - 1)) == 0 && uamd_tuning_get(UAMD_TUNE_GEMM_S) != 0) {
        bool whole = true;                      // ... and C rows that take 16-byte stores
        for (int i = 0; i < n_groups; ++i)
            whole = whole && (groups[i].N & (TN - 1)) == 0 && (groups[i].ldc & 7) == 0 && aligned16(groups[i].C) &&
                    (groups[i].bias == nullptr || (reinterpret_cast<uintptr_t>(groups[i].bias) & 7) == 0);
        if (whole) {
            if (dtype == UAMD_BF16) return bnn ? launch256s<bf16_t, true>(a, st) : launch256s<bf16_t, false>(a, st);
            if (dtype == UAMD_F16) return bnn ? launch256s<f16_t, true>(a, st) : launch256s<f16_t, false>(a, st);
            return UAMD_ERR_DTYPE;
        }
    }
    // persistent walk (UAMD_TUNE_GEMM_PERSIST: 1 = when every CU gets >= 4 tiles (default), 2 = whenever it gets more than
    // one, 0 = never). Measured (profiles/r02i_gemm_persist_ab.txt): +0.5 % at 7 tiles per CU, -2.5 % .. 0 at 2 tiles per CU
    // (static assignment loses the dispatcher's balancing) -- the kernel is power-limited, idle slots it removes come back
    // as clock.
    const int n_cu = cu_count();
    const int persist = uamd_tuning_get(UAMD_TUNE_GEMM_PERSIST);
    if (persist && K >= 4 * TK && (n_cu & 7) == 0 && a.total_tiles > n_cu && (persist >= 2 || a.total_tiles >= 4 * n_cu)) {
        if (dtype == UAMD_BF16) return bnn ? launch256p<bf16_t, true>(a, st, n_cu) : launch256p<bf16_t, false>(a, st, n_cu);
        if (dtype == UAMD_F16) return bnn ? launch256p<f16_t, true>(a, st, n_cu) : launch256p<f16_t, false>(a, st, n_cu);
        return UAMD_ERR_DTYPE;
    }
    if (dtype == UAMD_BF16) return bnn ? launch256<bf16_t, true>(a, st) : launch256<bf16_t, false>(a, st);
    if (dtype == UAMD_F16) return bnn ? launch256<f16_t, true>(a, st) : launch256<f16_t, false>(a, st);
    return UAMD_ERR_DTYPE;
}

extern "C" int uamd_gemm_nt_256(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                                int n_groups, int accumulate, int dtype, void* stream) {
    return gemm256_entry(A, lda, M, K, groups, n_groups, accumulate, dtype, stream, false);
}

// C_g[M, N_g] (+)= A[M, K] @ B_g[K, N_g] (+ rank block XK[M, Rk] @ BK_g[Rk, N_g]): B and BK row-major with the OUTPUT
// dimension contiguous. This is how the backward multiplies by a weight stored [out, in]: dX = dY @ W contracts
// over W's rows, so it reads the same row-major NF4 decode as the forward (fast_lora.py:156, :193-204).
extern "C" int uamd_gemm_nn_256(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                                int n_groups, int accumulate, int dtype, void* stream) {
    return gemm256_entry(A, lda, M, K, groups, n_groups, accumulate, dtype, stream, true);
}

// C_g[M, N_g] (+)= A[K, M]^T @ B_g[K, N_g]: both operands with the CONTRACTED dimension as rows. The weight gradient of
// a trainable dense projection, dW[out, in] (+)= dY[T, out]^T @ X[T, in] (what torch.nn.Linear's backward computes as
// grad_output.t().mm(input)); with accumulate != 0 it adds into the gradient buffer (gradient accumulation, and the
// chunked lm_head gradient of the fused linear-CE path). M % 8 == 0, N_g % 8 == 0, K % 64 == 0; no rank block.
extern "C" int uamd_gemm_tn_256(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                                int n_groups, int accumulate, int dtype, void* stream) {
    return gemm256_entry(A, lda, M, K, groups, n_groups, accumulate, dtype, stream, true, true);
}
