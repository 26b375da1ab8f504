// LoRA gradient products for gfx950: the rank-r "TN" contractions over the token dimension.
//
// Replaces, in the backward of the reference's manual-autograd blocks (unsloth/kernels/fast_lora.py:172-189,
// :476-495, :632-637), the twelve `addmm_` / `matmul` calls per decoder layer of the form
//     d_A = s * (dY @ B)^T-ish @ X        and        d_B = s * (X @ A^T)^T @ dY
// i.e.  G[r, n] = s * sum_m P[m, r] * Z[m, n]   with P = dY·B or X·A^T ([M, r], r <= 16 per problem, fp32 from
// uamd_lora_xa, rounded to the activation dtype on load exactly where the reference holds a bf16 tensor) and
// Z = X, dY, h, df, de ... ([M, N] activations, N = 1024..14336).
//
// These are memory-bound streaming passes over Z (2 B per 2*r flops). Two VALU versions were measured first
// (v_dot2c_f32_bf16, then v_pk_fma_f32 with 16-byte loads): both stalled near 3 TB/s because fp32 FMA throughput
// (64 flop/clk/SIMD on gfx950, packed or not) is only just the HBM rate for r = 16. This version does the
// contraction on the matrix cores; the contraction index being the ROW index of Z, the B operand comes from a
// row-major LDS tile through the transposing read ds_read_b64_tr_b16. Up to 8 problems (all products of one
// autograd Function) go in ONE launch. Split over the token dimension is deterministic: every wave owns one
// (128-column slab, row chunk) unit and writes one partial slab; a second tiny kernel sums the partials in fixed
// order.
#include "common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

#define UAMD_TN_MAX_PROBLEMS 8

namespace {

constexpr int TN_COLS = 128;          // columns per slab
constexpr int TN_R = 16;              // ranks per problem

struct TnArgs {
    int n_probs;
    int M;
    int S;                                   // number of row chunks (partials per problem)
    int rows_per_wave;                       // 128 / 256 / 512: one wave = one (slab, row chunk) unit
    int total_slabs;
    int slab_start[UAMD_TN_MAX_PROBLEMS + 1];
    int64_t ws_off[UAMD_TN_MAX_PROBLEMS];    // float offset of the problem's partials in `ws`
    float* ws;
    uamd_lora_tn_problem p[UAMD_TN_MAX_PROBLEMS];
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_t> {
    typedef f16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// One wave = one unit (128-column slab, row chunk), no block-level synchronisation. Per 32-row step:
//   * the Z tile [32 rows x 256 B] and the P tile [32 rows x 16 ranks fp32] arrive in LDS by LDS-DMA (10 wave
//     instructions), double-buffered, retired by a counted vmcnt: every byte this kernel reads from global memory
//     is in flight while the previous step computes, with no VGPRs held for it;
//   * G[r][n] += sum_m P[m][r] Z[m][n] is 8 v_mfma_f32_16x16x32 (i = rank, j = column, k = row): the A operand
//     is P^T (8 strided LDS reads, rounded to the activation dtype where the reference holds a bf16 tensor),
//     the B operand is Z with the contraction index on ROWS, which is what ds_read_b64_tr_b16 delivers from the
//     row-major tile (32-byte granule ^= (row & 3) | ((row >> 3) & 1) << 2 keeps the 8 rows of a 32-lane half on
//     8 different granules of the 256-byte bank row; applied on the DMA source address).
constexpr int TN_ZT = 32 * 256;                  // Z tile bytes
constexpr int TN_PT = 32 * 64;                   // P tile bytes
constexpr int TN_STAGE = TN_ZT + TN_PT;          // 10 KiB
// Stages of the per-wave ring: 2 (20 KiB per wave, two 4-wave blocks per CU = 8 waves x ONE stage in flight = 80 KiB per CU).
// Round 5 built the 4-stage ring (40 KiB per wave, one block per CU, THREE stages = 120 KiB per CU in flight,
// -DUAMD_TN_STAGES=4) on the hypothesis that the kernel is bound by bytes in flight: 184.9 vs 184.7 us on a layer's six MLP
// problems (906 MB, 4.9 TB/s), 66.5 vs 60.8 us on one [8192, 14336] problem (profiles/r05_lora_tn_stages_ab.txt) -- it is not;
// 4.9 TB/s is what workgroups that walk down rows in 256-byte pieces get from this memory system whatever they keep in
// flight (tools/probes/tile_shape_probe.hip, the note in front of launch_xa in glu.hip).
#ifndef UAMD_TN_STAGES
#define UAMD_TN_STAGES 2
#endif
#ifndef UAMD_TN_RPW_MAX
#define UAMD_TN_RPW_MAX 512       // most rows of one wave (fewer, longer waves = fewer fp32 partials to write and reduce)
#endif
#ifndef UAMD_TN_MIN_WAVES
#define UAMD_TN_MIN_WAVES 4096    // ... as long as the launch keeps this many waves (4 per SIMD)
#endif
constexpr int TN_NST = UAMD_TN_STAGES;
static_assert(TN_NST >= 2 && TN_NST <= 4, "ring of 2..4 stages (4 x 4 x 10 KiB = the 160 KiB of a CU)");
constexpr int TN_WAVE_LDS = TN_NST * TN_STAGE;   // 20 KiB per wave, 80 KiB per block

__device__ __forceinline__ void tn_dma(const void* gptr, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(lds_dst)
        : "memory");
}

template <typename T>
__global__ void __launch_bounds__(256) lora_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma16<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= (int64_t)a.total_slabs * a.S) return;          // wave-uniform; no block-level sync below
    const int slab_lin = (int)(unit % a.total_slabs);
    const int sblk = (int)(unit / a.total_slabs);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < UAMD_TN_MAX_PROBLEMS; ++i)
        if (i < a.n_probs && slab_lin >= a.slab_start[i]) pi = i;
    const uamd_lora_tn_problem& pr = a.p[pi];
    const int slab = slab_lin - a.slab_start[pi];
    const int n_slabs = a.slab_start[pi + 1] - a.slab_start[pi];
    const int M = a.M, N = pr.N, R = pr.R;
    const T* Z = (const T*)pr.Z;
    const float* P = pr.P;
    unsigned char* my = smem + wave * TN_WAVE_LDS;
    const unsigned my_lds = (unsigned)(uintptr_t)(lds_u8*)my;

    // ---- DMA plan. Z: instruction i (0..7) = rows 4i + (lane>>4), stored 16-B unit lane&15 of the 256-B row;
    //      the stored 32-B granule (unit>>1) holds logical granule ^ f(row). Columns past N re-read the last
    //      valid 16 bytes (their sums land in the padding of the partial slab and are never reduced).
    //      P: instruction i (0..1) = rows 16i + (lane>>2), 4 ranks at 4 (lane&3); ranks >= R re-read rank 0.
    const int zr4 = lane >> 4;
    int zcol[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + zr4;
        const int f = (row & 3) | (((row >> 3) & 1) << 2);
        const int u16 = ((((lane & 15) >> 1) ^ f) << 1) | (lane & 1);
        int col = slab * TN_COLS + u16 * 8;
        if (col + 8 > N) col = N - 8;
        zcol[i] = col;
    }
    const int prow = lane >> 2;
    const int pc = (lane & 3) * 4 + 4 <= ((R + 3) & ~3) ? (lane & 3) * 4 : 0;
    auto issue = [&](int m0, int stage) {
        const unsigned d = my_lds + stage * TN_STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int row = m0 + 4 * i + zr4;
            row = row < M ? row : M - 1;
            tn_dma(Z + (int64_t)row * pr.ldz + zcol[i], d + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = m0 + 16 * i + prow;
            row = row < M ? row : M - 1;
            tn_dma(P + (int64_t)row * pr.ldp + pc, d + TN_ZT + i * 1024);
        }
    };

    // ---- per-lane LDS read addresses
    const int l15 = lane & 15, g4 = lane >> 4;
    const int sg2 = l15 >> 2;
    const int f_rd = sg2 | ((g4 & 1) << 2);
    // transposing reads of Z: row 8 g4 + sg2 (+4 for the second read), granule jt ^ f, 8 B at 8 (l15 & 3)
    const int z_lane = (8 * g4 + sg2) * 256 + f_rd * 32 + (l15 & 3) * 8;
    // P^T: rank l15, rows 8 g4 + e
    const int p_lane = TN_ZT + (8 * g4) * 64 + l15 * 4;

    f32x4_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int m_base = sblk * a.rows_per_wave;
    const int m_end = min(m_base + a.rows_per_wave, M);
    const int nch = (m_end - m_base + 31) / 32;
#pragma unroll
    for (int i = 0; i < TN_NST - 1; ++i)
        if (i < nch) issue(m_base + 32 * i, i);
    for (int ch = 0; ch < nch; ++ch) {
        const int m0 = m_base + ch * 32;
        const int stage = ch % TN_NST;
        // the stage (ch - 1) % NST was read in the previous step (its ds_reads have returned: the MFMAs consumed them): refill
        // it with the step NST - 1 ahead, then wait until only the NST - 1 younger stages (10 DMA instructions each) are out
        if (ch + TN_NST - 1 < nch) {
            issue(m0 + 32 * (TN_NST - 1), (ch + TN_NST - 1) % TN_NST);
            if (TN_NST == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (TN_NST == 3) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        } else {
            const int younger = nch - 1 - ch;            // stages still in flight behind this one: 0 .. NST - 2
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* sz = my + stage * TN_STAGE;
        // A operand: P^T[rank][rows], rounded to T; rows past M and ranks past R contribute zero
        union { T e[8]; frag_t f; } pa;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = *reinterpret_cast<const float*>(sz + p_lane + e * 64);
            if (m0 + 8 * g4 + e >= M || l15 >= R) x = 0.f;
            pa.e[e] = from_f32<T>(x);
        }
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            union { s16x4_t h[2]; frag_t f; } zb;
            const int a0 = z_lane ^ (jt * 32);
            zb.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sz + a0));
            zb.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sz + a0 + 4 * 256));
            acc[jt] = Mfma16<T>::run(pa.f, zb.f, acc[jt]);
        }
    }

    // ---- partial store: part[sblk][r][n], n padded to whole slabs. C layout: column = lane & 15, rank = 4 (lane>>4) + reg
    const int64_t npad = (int64_t)n_slabs * TN_COLS;
    float* part = a.ws + a.ws_off[pi] + (int64_t)sblk * TN_R * npad + slab * TN_COLS + l15;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(int64_t)(4 * g4 + r) * npad + jt * 16] = acc[jt][r];
}

// out = scale * sum_s part[s]; out_nr == 0: out[r * ldo + n], else out[n * ldo + r]
__global__ void __launch_bounds__(256) lora_tn_reduce_kernel(TnArgs a) {
    const int pi = blockIdx.y;
    const uamd_lora_tn_problem& pr = a.p[pi];
    const int n_slabs = a.slab_start[pi + 1] - a.slab_start[pi];
    const int64_t npad = (int64_t)n_slabs * TN_COLS;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over R x N
    const int N = pr.N, R = pr.R;
    if (idx >= (int64_t)R * N) return;
    const int r = (int)(idx / N), n = (int)(idx - (int64_t)r * N);
    const float* part = a.ws + a.ws_off[pi] + (int64_t)r * npad + n;
    // same summation order as a plain loop, but 8 loads in flight per round instead of one load -> wait -> add
    float v = 0.f;
    const int64_t sstride = (int64_t)TN_R * npad;
    int s = 0;
    for (; s + 8 <= a.S; s += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(s + u) * sstride];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < a.S; ++s) v += part[(int64_t)s * sstride];
    v *= pr.scale;
    float* o = (pr.out_nr & 1) ? pr.out + (int64_t)n * pr.ldo + r : pr.out + (int64_t)r * pr.ldo + n;
    *o = (pr.out_nr & 2) ? *o + v : v;                 // bit 1: accumulate into `out` (gradient arena)
}

// ------------------------------------------------------------------------------------------------------------
// XA[M, R] = X[M, K] @ W[R, K]^T (fp32 out): the skinny LoRA product of the forward (X @ A^T) and of the
// backward (dY @ B). Replaces the first version in gemm.hip (16 rows per block, fragments straight from
// global): that one re-read all of W per 16 rows (L2 traffic 3x the HBM traffic at R = 48: 1.3 TB/s) and kept
// only a few KB in flight per CU.
// Here a block owns 32 rows; its 4 waves split K four ways and each streams its [32 rows x K/4] slab of X and the
// matching slab of W through a private double-buffered LDS ring by LDS-DMA (64-deep steps, counted vmcnt, no
// block-level synchronisation until the end), multiplies with v_mfma_f32_16x16x32 (both operands K-contiguous,
// plain row reads, 16-byte slot ^= (row >> 1) & 7 against bank conflicts), and the four partial [32 x R] tiles
// are summed in fixed order through LDS.
constexpr int XA_KS = 64;                              // k per step
constexpr int XA_XT = 32 * XA_KS * 2;                  // X tile bytes (4 KiB)
template <int NT> struct XaCfg {
    static constexpr int WT = NT * 16 * XA_KS * 2;     // W tile bytes (2 KiB per 16 ranks)
    static constexpr int STAGE = XA_XT + WT;
    static constexpr int WAVE = 2 * STAGE;
    static constexpr int RED = 4 * 32 * NT * 16 * 4;   // final reduction buffer
    static constexpr int LDS = 4 * WAVE > RED ? 4 * WAVE : RED;
};

template <typename T, int NT>
__global__ void __launch_bounds__(256) lora_xa2_kernel(const T* __restrict__ X, int64_t ldx,
                                                       const T* __restrict__ W, int64_t ldw,
                                                       float* __restrict__ out, int64_t ld_out,
                                                       T* __restrict__ out_k, int64_t ld_k, int k_cols,
                                                       int M, int K, int R, int out_cols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma16<T>::frag frag_t;
    typedef XaCfg<NT> C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int m0 = blockIdx.x * 32;
    unsigned char* my = smem + wave * C::WAVE;
    const unsigned my_lds = (unsigned)(uintptr_t)(lds_u8*)my;

    // this wave's K range: whole 64-steps, split as evenly as possible; the ragged K tail (K % 64) goes to the
    // last wave and is handled by clamping + zeroing
    const int nsteps_all = (K + XA_KS - 1) / XA_KS;
    const int s_beg = (nsteps_all * wave) / 4, s_end = (nsteps_all * (wave + 1)) / 4;

    // DMA plan. X: instruction i (0..3) = rows 8i + (lane>>3), stored 16-B slot lane&7 holds logical slot
    // (lane&7) ^ ((row>>1)&7). W: instruction i (0..2NT-1) = ranks 8i + (lane>>3), same swizzle.
    const int drow = lane >> 3;
    int xoff[4], woff[2 * NT];
    int xslot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + drow;
        xslot[i] = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        xoff[i] = gm;                                        // row index; column added per step
    }
    int wslot[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
        const int row = 8 * i + drow;
        wslot[i] = (lane & 7) ^ ((row >> 1) & 7);
        woff[i] = row < R ? row : R - 1;
    }
    auto issue = [&](int step, int stage) {
        const unsigned d = my_lds + stage * C::STAGE;
        const int k0 = step * XA_KS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int k = k0 + xslot[i] * 8;
            k = k + 8 <= K ? k : K - 8;                      // ragged tail: valid bytes, zeroed at use
            tn_dma(X + (int64_t)xoff[i] * ldx + k, d + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            int k = k0 + wslot[i] * 8;
            k = k + 8 <= K ? k : K - 8;
            tn_dma(W + (int64_t)woff[i] * ldw + k, d + XA_XT + i * 1024);
        }
    };

    f32x4_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read: row l15 (+16 per tile), logical slot 4 ks + g4 -> stored slot ^ ((row>>1)&7); rows 16..31
    // have the same (row>>1)&7 pattern shifted: (row>>1)&7 = ((l15>>1) + 8*tile)&7 = (l15>>1)&7
    const int sw = (l15 >> 1) & 7;
    const int f_lane = l15 * 128;
    constexpr int NDMA = 4 + 2 * NT;
    // every block reads the SAME W slab; blocks start at rotated positions of their K range so that 256 CUs do not
    // hammer the same L2 lines in the same microsecond (fixed per block: results stay run-to-run identical)
    const int nst = s_end - s_beg;
    const int rot = nst > 0 ? (int)(blockIdx.x % nst) : 0;
    auto step_of = [&](int i) { const int j = i + rot; return s_beg + (j >= nst ? j - nst : j); };
    if (nst > 0) issue(step_of(0), 0);
    for (int it = 0; it < nst; ++it) {
        const int st = step_of(it);
        const int stage = it & 1;
        if (it + 1 < nst) {
            issue(step_of(it + 1), stage ^ 1);
            if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (NDMA == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* sx = my + stage * C::STAGE;
        const unsigned char* swt = sx + XA_XT;
        const bool tail = st * XA_KS + XA_KS > K;             // wave-uniform
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((4 * ks + g4) ^ sw) * 16;
            union FU { uint4 r; frag_t f; };
            FU xa[2], wb[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) xa[i].r = *reinterpret_cast<const uint4*>(sx + i * 2048 + f_lane + slot);
#pragma unroll
            for (int j = 0; j < NT; ++j) wb[j].r = *reinterpret_cast<const uint4*>(swt + j * 2048 + f_lane + slot);
            if (tail) {
                // columns past K: the DMA re-read the last 16 valid bytes; contribute nothing
                if (st * XA_KS + ks * 32 + g4 * 8 + 8 > K) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) xa[i].r = make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = Mfma16<T>::run(xa[i].f, wb[j].f, acc[i][j]);
        }
    }

    // ---- fixed-order reduction of the 4 K-quarters; C layout: column (rank) = l15, row = 4 g4 + reg
    __syncthreads();                                          // every wave is done with its DMA ring
    float* red = reinterpret_cast<float*>(smem);              // [4][32 rows][NT*16]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 32 + i * 16 + 4 * g4 + r) * (NT * 16) + j * 16 + l15] = acc[i][j][r];
    __syncthreads();
    if (out != nullptr) {
        for (int idx = tid; idx < 32 * out_cols; idx += 256) {
            const int mm = idx / out_cols, c = idx - mm * out_cols;
            if (m0 + mm < M) {
                float v = 0.f;
                if (c < R) {
                    const float* q = red + mm * (NT * 16) + c;
                    v = ((q[0] + q[32 * NT * 16]) + q[2 * 32 * NT * 16]) + q[3 * 32 * NT * 16];
                }
                out[(int64_t)(m0 + mm) * ld_out + c] = v;          // columns R..out_cols-1 are zero padding
            }
        }
    }
    // the same sums rounded to the activation dtype (where the reference holds `X @ A.to(dtype)`, utils.py:1166),
    // zero-padded to k_cols: the rank block the 256x256 GEMM contracts as extra K tiles. 8 columns per thread.
    if (out_k != nullptr) {
        const int vec_per_row = k_cols >> 3;
        for (int idx = tid; idx < 32 * vec_per_row; idx += 256) {
            const int mm = idx / vec_per_row, c0 = (idx - mm * vec_per_row) * 8;
            if (m0 + mm < M) {
                Vec16<T> v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + j;
                    float x = 0.f;
                    if (c < R) {
                        const float* q = red + mm * (NT * 16) + c;
                        x = ((q[0] + q[32 * NT * 16]) + q[2 * 32 * NT * 16]) + q[3 * 32 * NT * 16];
                    }
                    v.e[j] = from_f32<T>(x);
                }
                st16(out_k + (int64_t)(m0 + mm) * ld_k + c0, v);
            }
        }
    }
}

template <typename T, int NT>
int launch_xa2(const void* X, int64_t ldx, const void* W, int64_t ldw, float* out, int64_t ld_out, void* out_k,
               int64_t ld_k, int k_cols, int M, int K, int R, int out_cols, hipStream_t st) {
    static bool done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lora_xa2_kernel<T, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, XaCfg<NT>::LDS);
        if (e != hipSuccess) return (int)e;
        done[dev] = true;
    }
    hipLaunchKernelGGL((lora_xa2_kernel<T, NT>), dim3((unsigned)((M + 31) / 32)), dim3(256), XaCfg<NT>::LDS, st,
                       (const T*)X, ldx, (const T*)W, ldw, out, ld_out, (T*)out_k, ld_k, k_cols, M, K, R, out_cols);
    return uamd_launch_status();
}

template <typename T>
int xa2_dispatch(const void* X, int64_t ldx, const void* W, int64_t ldw, float* out, int64_t ld_out, void* out_k,
                 int64_t ld_k, int k_cols, int M, int K, int R, int out_cols, hipStream_t st) {
    const int nt = (R + 15) / 16;
    if (nt <= 1) return launch_xa2<T, 1>(X, ldx, W, ldw, out, ld_out, out_k, ld_k, k_cols, M, K, R, out_cols, st);
    if (nt <= 2) return launch_xa2<T, 2>(X, ldx, W, ldw, out, ld_out, out_k, ld_k, k_cols, M, K, R, out_cols, st);
    if (nt <= 3) return launch_xa2<T, 3>(X, ldx, W, ldw, out, ld_out, out_k, ld_k, k_cols, M, K, R, out_cols, st);
    if (nt <= 4) return launch_xa2<T, 4>(X, ldx, W, ldw, out, ld_out, out_k, ld_k, k_cols, M, K, R, out_cols, st);
    return UAMD_ERR_ARG;
}

template <typename K_>
int tn_set_attr(K_ kernel, bool* done) {
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TN_WAVE_LDS);
        if (e != hipSuccess) return (int)e;
        *done = true;
    }
    return 0;
}

}  // namespace

extern "C" int uamd_lora_tn(const uamd_lora_tn_problem* probs, int n_probs, int M, float* workspace,
                            int64_t workspace_floats, int dtype, void* stream) {
    if (!probs || n_probs < 1 || n_probs > UAMD_TN_MAX_PROBLEMS || M < 0 || !workspace) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    TnArgs a;
    a.n_probs = n_probs; a.M = M; a.ws = workspace;
    {   // rows per wave: as many as keep >= 4096 waves in the launch (4 per SIMD), fewer partials otherwise
        int64_t slabs_all = 0;
        for (int i = 0; i < n_probs; ++i) slabs_all += (probs[i].N + TN_COLS - 1) / TN_COLS;
        int rpw = UAMD_TN_RPW_MAX;
        while (rpw > 128 && slabs_all * ((M + rpw - 1) / rpw) < UAMD_TN_MIN_WAVES) rpw >>= 1;
        a.rows_per_wave = rpw;
        a.S = (M + rpw - 1) / rpw;
    }
    int slabs = 0;
    int64_t off = 0;
    int max_rn_blocks = 1;
    for (int i = 0; i < UAMD_TN_MAX_PROBLEMS; ++i) {
        a.slab_start[i] = slabs;
        if (i < n_probs) {
            const uamd_lora_tn_problem& p = probs[i];
            if (!p.P || !p.Z || !p.out || p.N <= 0 || p.R <= 0 || p.R > TN_R) return UAMD_ERR_ARG;
            if ((p.N & 7) || p.N < 8 || (p.ldz & 7) || (p.ldp & 3) || !aligned16(p.Z) || !aligned16(p.P)) return UAMD_ERR_ALIGN;
            a.p[i] = p;
            a.ws_off[i] = off;
            const int ns = (p.N + TN_COLS - 1) / TN_COLS;
            slabs += ns;
            off += (int64_t)a.S * TN_R * ns * TN_COLS;
            const int64_t rn = ((int64_t)p.R * p.N + 255) / 256;
            if (rn > max_rn_blocks) max_rn_blocks = (int)rn;
        } else {
            a.p[i] = probs[0];
            a.ws_off[i] = 0;
        }
    }
    a.slab_start[UAMD_TN_MAX_PROBLEMS] = slabs;
    for (int i = n_probs; i < UAMD_TN_MAX_PROBLEMS; ++i) a.slab_start[i + 1] = slabs;
    a.total_slabs = slabs;
    if (off > workspace_floats) return UAMD_ERR_ARG;
    const int64_t blocks = ((int64_t)slabs * a.S + 3) / 4;
    if (blocks > 0x7fffffffLL) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[2][64] = {{false}};
    int dev = 0, rc;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (dtype == UAMD_BF16) {
        if ((rc = tn_set_attr(&lora_tn_kernel<bf16_t>, &attr_set[0][dev]))) return rc;
        hipLaunchKernelGGL((lora_tn_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 4 * TN_WAVE_LDS, st, a);
    } else if (dtype == UAMD_F16) {
        if ((rc = tn_set_attr(&lora_tn_kernel<f16_t>, &attr_set[1][dev]))) return rc;
        hipLaunchKernelGGL((lora_tn_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 4 * TN_WAVE_LDS, st, a);
    } else {
        return UAMD_ERR_DTYPE;
    }
    rc = uamd_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(lora_tn_reduce_kernel, dim3((unsigned)max_rn_blocks, (unsigned)n_probs), dim3(256), 0, st, a);
    return uamd_launch_status();
}

// XA[M, out_cols] = X[M, K] @ W[R, K]^T in fp32 (columns >= R zero-filled), streaming version for R <= 64.
extern "C" int uamd_lora_xa2k(const void* X, int64_t ldx, const void* W, int64_t ldw, float* out, int64_t ld_out,
                              void* out_k, int64_t ld_k, int k_cols, int M, int K, int R, int out_cols, int dtype,
                              void* stream) {
    if (M < 0 || K < 8 || R <= 0 || R > 64) return UAMD_ERR_ARG;
    if (!out && !out_k) return UAMD_ERR_ARG;
    if (out && (out_cols < R || out_cols > 256)) return UAMD_ERR_ARG;
    if (out_k && (k_cols < R || k_cols > 256)) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 7) || (ldx & 7) || (ldw & 7) || !aligned16(X) || !aligned16(W)) return UAMD_ERR_ALIGN;
    if (out_k && ((k_cols & 7) || (ld_k & 7) || !aligned16(out_k))) return UAMD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16)
        return xa2_dispatch<bf16_t>(X, ldx, W, ldw, out, ld_out, out_k, ld_k, k_cols, M, K, R, out_cols, st);
    if (dtype == UAMD_F16)
        return xa2_dispatch<f16_t>(X, ldx, W, ldw, out, ld_out, out_k, ld_k, k_cols, M, K, R, out_cols, st);
    return UAMD_ERR_DTYPE;
}

extern "C" int uamd_lora_xa2(const void* X, int64_t ldx, const void* W, int64_t ldw, float* out, int64_t ld_out,
                             int M, int K, int R, int out_cols, int dtype, void* stream) {
    if (!out) return UAMD_ERR_ARG;
    return uamd_lora_xa2k(X, ldx, W, ldw, out, ld_out, nullptr, 0, 0, M, K, R, out_cols, dtype, stream);
}

// ------------------------------------------------------------------------------------------------------------
// Once per optimizer step: activation-dtype copies of every LoRA factor, row-major AND transposed, in ONE launch.
// The reference casts at every use (`A.to(dtype)`, `B.to(dtype)`, unsloth/kernels/utils.py:1166-1167,
// fast_lora.py:138-145), i.e. ~1,400 tiny cast / transpose kernels per step at 7 projections x 32 layers.
// Descriptor table lives in device memory (built once per model); block -> matrix by binary search over the
// prefix sum of 32x32 tiles.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) lora_prepare_kernel(const uamd_lora_prep_desc* __restrict__ d, int n_mats,
                                                          const int* __restrict__ tile_prefix) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_mats - 1;                       // last matrix whose first tile <= blockIdx.x
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_prefix[mid] <= bid) lo = mid; else hi = mid - 1;
    }
    const uamd_lora_prep_desc m = d[lo];
    const int t = bid - tile_prefix[lo];
    const int tiles_c = (m.cols + 31) / 32;
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8
    const float* src = (const float*)m.src;
    T* rm = (T*)m.dst_rowmajor;
    T* tr = (T*)m.dst_transposed;
    T* pad = (T*)m.dst_pad;                                          // scaled copy with its own leading dimension
    const bool pad_t = m.pad_transposed != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        float v = 0.f;
        if (r < m.rows && c < m.cols) {
            v = src[(int64_t)r * m.cols + c];
            if (rm) rm[(int64_t)r * m.cols + c] = from_f32<T>(v);
            if (pad && !pad_t) pad[(int64_t)r * m.pad_ld + c] = from_f32<T>(m.pad_scale * v);
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
    if (tr || (pad && pad_t)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;              // transposed: [cols][rows]
            if (r < m.rows && c < m.cols) {
                const float v = tile[tx][ty + 8 * i];
                if (tr) tr[(int64_t)c * m.rows + r] = from_f32<T>(v);
                if (pad && pad_t) pad[(int64_t)c * m.pad_ld + r] = from_f32<T>(m.pad_scale * v);
            }
        }
    }
}
}  // namespace

extern "C" int uamd_lora_prepare(const uamd_lora_prep_desc* descs_dev, const int* tile_prefix_dev, int n_mats,
                                 int total_tiles, int dtype, void* stream) {
    if (!descs_dev || !tile_prefix_dev || n_mats <= 0 || total_tiles < 0) return UAMD_ERR_ARG;
    if (total_tiles == 0) return UAMD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16)
        hipLaunchKernelGGL((lora_prepare_kernel<bf16_t>), dim3((unsigned)total_tiles), dim3(256), 0, st, descs_dev, n_mats, tile_prefix_dev);
    else if (dtype == UAMD_F16)
        hipLaunchKernelGGL((lora_prepare_kernel<f16_t>), dim3((unsigned)total_tiles), dim3(256), 0, st, descs_dev, n_mats, tile_prefix_dev);
    else
        return UAMD_ERR_DTYPE;
    return uamd_launch_status();
}
