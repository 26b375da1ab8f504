// Causal GQA flash attention for gfx950, head_dim 128, bf16/fp16 — forward.
//
// Where it sits on the reference's path: the step between RoPE and o_proj, `run_attention`
// (unsloth/utils/attention_dispatch.py:298-617) called from LlamaAttention_fast_forward
// (unsloth/models/llama.py:671-770), which hands Q/K/V to flash-attn / xformers / torch SDPA. SURVEY 8(f1):
// first "next" component. Semantics = softmax(Q K^T / sqrt(d) + causal mask) V per (batch, head), GQA by head
// index (q head h reads kv head h / (Hq/Hk)), fp32 softmax statistics, one rounding of P to the activation
// dtype before P·V (what flash-attention and SDPA do), log-sum-exp kept for the backward.
//
// CDNA4 design.
//   * Layout: Q/K/V are read where the QKV GEMM left them, [B, T, H, D] with arbitrary element strides (d
//     contiguous), and O is written as [B, T, Hq*D]: no [B,H,T,D] transposes or .contiguous() copies on
//     either side (the reference pays them, llama.py:276-277, :757).
//   * One block = 8 waves = one (batch, KV head, tile of q positions): wave w handles q head kvh*G + w%G and
//     32 q positions, so the G query heads of a GQA group share every K/V tile in LDS (K/V traffic / G).
//   * K/V tiles (64 keys x 128 d, 16 KiB each) go HBM/L2 -> LDS by LDS-DMA in whole 256-byte rows, 3-stage
//     ring, one barrier per tile, counted vmcnt (a tile in flight across every barrier).
//   * S^T = K Q^T with v_mfma_f32_32x32x16: the C layout then has the q position on the LANE, so the softmax
//     statistics (running max, sum, rescale factor) are per-lane scalars, and the 8 registers of one k-step
//     ARE the B operand of O^T += V^T P^T after a cvt to bf16: no LDS round trip, no cross-lane shuffle for P.
//     The matching V^T operand comes straight from the row-major V tile with ds_read_b64_tr_b16 (probed
//     semantics: profiles/r01_tr_probe.txt), so V is never transposed in memory.
//   * Bank conflicts: K rows are read 16 bytes per lane down a column -> 16-byte slot ^= row & 15; V rows are
//     read by the transposing 8-byte reads, 4 rows x 64 B per 32 lanes -> slot ^= (row & 3) << 2. Both swizzles
//     are applied on the per-lane DMA SOURCE address (the DMA destination is lane-linear).
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

namespace {

template <typename T> struct MfmaA;
template <> struct MfmaA<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct MfmaA<f16_t> {
    typedef f16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int AD = 128;                  // head dim
constexpr int KT = 64;                   // keys per tile
constexpr int TILE_B = KT * AD * 2;      // 16 KiB
constexpr int STAGE_B = 2 * TILE_B;      // K + V
constexpr int NST = 3;
constexpr int ATTN_LDS = NST * STAGE_B;  // 96 KiB

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4a_t;
typedef __attribute__((address_space(3))) u32x4a_t lds_u32x4a;
typedef __attribute__((ext_vector_type(2))) float f32x2a_t;
typedef __attribute__((address_space(3))) f32x2a_t lds_f32x2a;

struct AttnArgs {
    const void* Q; const void* K; const void* V; void* O; float* LSE;
    const int* lo;                       // [B, T] first key each query may attend (NULL: 0), non-decreasing in t
    const int* hi;                       // [B, T] LAST key each query may attend, >= t, non-decreasing (NULL: t itself = causal).
                                         // Non-causal (bidirectional) attention inside documents: lo = document start, hi = its end
    int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
    int B, T, Hq, Hk, G, nsub;          // G = query heads per block (1, 2, 4, 8), nsub = 8 / G q-subtiles of 32 rows per block
    int kvm;                             // Hk counts VIRTUAL KV heads: virtual head v reads K / V head v / kvm and serves query heads
                                         // v G .. v G + G - 1 (group sizes 3, 5, 6, 7 = kvm x the largest of 1, 2, 4, 8 dividing them)
    int nqt;                             // number of q tiles
    int D;                               // head dim, a multiple of 8, <= 128. Below 128 (attn_fwd_kernel only) the kernel still works
                                         // on 256-byte rows: LDS-DMA lanes whose 16-byte slot lies past D re-read slot 0 of their row
                                         // (in bounds, finite), the Q fragments past D are zero registers -- so Q K^T sees only the real
                                         // columns -- and the rows of O^T past D are never stored. No padded copies of Q / K / V.
    int lse_st;                          // row stride of LSE [B, Hq, lse_st] (T rounded up to 32)
    float scale_log2;                    // softmax scale * log2(e)
    int* ctr;                            // attn_fwd_ps_kernel<.., DYN = true>: {claim counter, finished workgroups}, both zero at launch
                                         // and zeroed again by the last workgroup to finish (attn_ctr_slot)
};

// Block index -> (rank of the tile in heaviest-first order, (batch, KV head) pair): tile-major, pair-minor. With 8 KV heads the
// pair index mod 8 -- the XCD a block is dispatched to -- is the KV head: an XCD's L2 serves one KV head of every batch row.
// Round 4 tried the opposite order (an XCD works through its pairs ONE AFTER THE OTHER, so that its 32 CUs share 1-2 MiB of
// K / V instead of 4): 6 % SLOWER in forward and backward (profiles/r04_attn_ab_xcd_map_and_pingpong.jsonl) -- 32 CUs asking
// for the same tiles at the same time queue on the same L2 channels.
__device__ __forceinline__ void block_to_work(int bid, int npairs, int& rank, int& pair) {
    rank = bid / npairs;
    pair = bid % npairs;
}

// two LDS-DMA wave-instructions (2 x 1 KiB) from one wave-uniform base: lane l copies 16 B from
// base + voff_i to LDS [dst_i + 16 l). Inline asm: see gemm256.hip / cdna guide 5.7.
__device__ __forceinline__ void dma16x2(const void* base, unsigned v0, unsigned v1, unsigned d0, unsigned d1) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "s"(d0), "s"(d1), "s"(base)
        : "memory");
}

__device__ __forceinline__ void dma16x4g(const void* base, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                         unsigned d0) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %6\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %6\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %6\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %6\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(d0), "s"(base)
        : "memory", "scc");
}
__device__ __forceinline__ void dma16x1(const void* gptr, unsigned d0) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(d0)
        : "memory");
}

// max over the two lane halves (lanes l and l ^ 32) without touching the LDS pipe: v_permlane32_swap is a VALU
// op, so the softmax does not wait (lgkmcnt) for operand reads that are in flight for the next MFMA phase.
__device__ __forceinline__ float max_across_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <typename T>
__device__ __forceinline__ uint32_t pack_pair(float lo, float hi) {
    union { T h[2]; uint32_t u; } v;
    v.h[0] = from_f32<T>(lo);
    v.h[1] = from_f32<T>(hi);
    return v.u;
}
// the same through a vector conversion: selects the single v_cvt_pk_{bf16,f16}_f32 (the two scalar conversions above cost
// 4 VALU instructions per pair in bf16). The round-1 code paths keep the form they were tuned with
// (the packed form moves their register allocation over the 256-register edge).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v_t;
template <typename T>
__device__ __forceinline__ uint32_t pack_pair2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    if constexpr (std::is_same<T, bf16_t>::value) {
        union { bf16x2v_t h; uint32_t u; } r;
        r.h = __builtin_convertvector(v, bf16x2v_t);
        return r.u;
    } else {
        union { f16x2v_t h; uint32_t u; } r;
        r.h = __builtin_convertvector(v, f16x2v_t);
        return r.u;
    }
}

// -DUAMD_ATTN_TRACE: s_memtime stamps around the phases of forward tiles 8 and 9 (tools/attn_trace.py); never in the
// shipped library.
#ifdef UAMD_ATTN_TRACE
__device__ unsigned* g_attn_trace = nullptr;
#define ASTAMP(TI, I)                                                                    \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        if ((TI) == 8 || (TI) == 9) {                                                    \
            ats[(((TI) & 1) << 3) + (I)] = (unsigned)__builtin_amdgcn_s_memtime();       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
        }                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define ASTAMP(TI, I) do { } while (0)
#endif

// Row-per-lane epilogue store of a 32 x 128 accumulator block (lane = row l31; lane half lh holds columns 8 g + 4 lh .. + 3
// of every 8-column group g): 16 x 8-byte stores per lane are store-ISSUE-bound (cdna_hip_programming.md T21). One
// v_permlane32_swap per dword of a group pair (g, g + 1) hands the lower half 16 contiguous bytes of group g and the upper
// half 16 of group g + 1: 8 x dwordx4 per lane, same bytes, same addresses.
template <typename T>
__device__ __forceinline__ void store_rows_x4(T* row, const f32x16_t (&acc)[4], float mul, int lh, bool live, int D = 128) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; qd += 2) {
            uint32_t ax = pack_pair2<T>(acc[dt][qd * 4 + 0] * mul, acc[dt][qd * 4 + 1] * mul);
            uint32_t ay = pack_pair2<T>(acc[dt][qd * 4 + 2] * mul, acc[dt][qd * 4 + 3] * mul);
            uint32_t bx = pack_pair2<T>(acc[dt][qd * 4 + 4] * mul, acc[dt][qd * 4 + 5] * mul);
            uint32_t by = pack_pair2<T>(acc[dt][qd * 4 + 6] * mul, acc[dt][qd * 4 + 7] * mul);
#ifdef UAMD_ATTN_NARROW_STORE
            if (live) {
                *reinterpret_cast<uint2*>(row + dt * 32 + qd * 8 + lh * 4) = make_uint2(ax, ay);
                *reinterpret_cast<uint2*>(row + dt * 32 + qd * 8 + 8 + lh * 4) = make_uint2(bx, by);
            }
#else
            const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            if (live && dt * 32 + qd * 8 + lh * 8 < D) *reinterpret_cast<uint4*>(row + dt * 32 + qd * 8 + lh * 8) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
#endif
        }
}

// BAND = false: plain causal attention, the band bookkeeping folds away at compile time (it costs ~50 VGPRs).
// (A 2-stage ring = 64 KiB = two blocks per CU was tried in round 4: the kernel needs ~240 registers per lane -- O 64, S 32, Q^T 32,
// operand fragments -- so the 128-register cap of 4 waves per SIMD spills 126-256 registers. Not built.)
// lgkmcnt(0) as the BUILTIN (an S_WAITCNT hipcc's own wait insertion sees and accounts for) behind a compiler-level memory
// barrier: behind an inline-asm wait the compiler still counts the reads as pending and puts a (satisfied) counted wait in
// front of every MFMA that takes one -- 21 instructions per tile step of the persistent forward.
__device__ __forceinline__ void wait_lgkm0() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("" ::: "memory");
}

// 16-byte slot `s` (8 elements) of a 256-byte row when the head has only D elements: slots past D re-read slot 0
__device__ __forceinline__ int slot_in(int s, int D) { return s * 8 < D ? s : 0; }

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }     // one v_max3_f32

template <typename T, bool BAND, int DC>
__global__ void __launch_bounds__(512, 2) attn_fwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 32 * p.nsub;
    // longest-processing-time-first over the WHOLE grid: all (batch, kv head) pairs of the heaviest q tile are
    // dispatched first, the one-tile blocks fill the tail
    const int npairs = p.Hk * p.B;
    int rank_, pair_;
    block_to_work((int)blockIdx.x, npairs, rank_, pair_);
    const int qtile = p.nqt - 1 - rank_;                              // heaviest q tiles first
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int head = kvh * G + (wave % G);
    const int qs = qtile * QT + (wave / G) * 32;                      // first q position of this wave
    const int q_pos = qs + l31;
    const int q_ld = q_pos < T_ ? q_pos : T_ - 1;
    // band lower edge (packed documents / sliding window): per-lane, and -- lo being non-decreasing -- lane 0 /
    // lane 31 give the wave's min / max, the block's first row the block's min
    const int lo_q = BAND ? p.lo[(int64_t)b * T_ + q_ld] : 0;
    const int lo_w0 = BAND ? __builtin_amdgcn_readfirstlane(lo_q) : 0, lo_w1 = BAND ? __builtin_amdgcn_readlane(lo_q, 31) : 0;
    const int t_first = BAND ? p.lo[(int64_t)b * T_ + min(qtile * QT, T_ - 1)] / KT : 0;
    // upper edge: the query itself (causal) or the band's `hi` (non-causal attention inside documents; BAND builds only)
    const bool full = BAND && p.hi != nullptr;
    const int lim_q = full ? p.hi[(int64_t)b * T_ + q_ld] : q_pos;
    const int lim_w0 = full ? __builtin_amdgcn_readfirstlane(lim_q) : qs;
    const int lim_w1 = full ? __builtin_amdgcn_readlane(lim_q, 31) : min(qs + 31, T_ - 1);
    const int lim_blk = full ? p.hi[(int64_t)b * T_ + min(qtile * QT + QT - 1, T_ - 1)] : qtile * QT + QT - 1;

    // ---- Q^T operand fragments (B operand: lane -> q = l31, 8 d at 16 ks + 8 lh), kept for the whole tile loop
    frag_t qf[8];
    {
        const T* qp = (const T*)p.Q + b * p.q_sb + (int64_t)q_ld * p.q_st + (int64_t)head * p.q_sh + lh * 8;
        // MASKED = head dims below 128: fragments past D are zero. The load stays UNCONDITIONAL -- a lane past D re-reads the row's
        // first 16 bytes -- and the zeroing is an AND with a lane mask, not an `if`: behind per-lane branches hipcc waits for each
        // load before it issues the next (the dQ kernel's 24 prologue loads one by one: +9 % on the whole kernel at head_dim 128,
        // profiles/r06zm_attn_bisect.txt). head_dim 128 (a wave-uniform test) keeps the plain loads.
        auto load_q = [&](auto masked_c) __attribute__((always_inline)) {
            constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bool in = !MASKED || ks * 16 + lh * 8 < p.D;
                union { uint4 r; frag_t f; } u;
                u.r = *reinterpret_cast<const uint4*>(in ? qp + ks * 16 : qp - lh * 8);
                if constexpr (MASKED) {
                    const unsigned keep = in ? 0xffffffffu : 0u;
                    u.r.x &= keep; u.r.y &= keep; u.r.z &= keep; u.r.w &= keep;
                }
                qf[ks] = u.f;
            }
        };
        if (DC == AD && p.D == AD) load_q(std::false_type{});
        else load_q(std::true_type{});
    }

    // ---- DMA plan: a stage = K tile (64 rows x 256 B) then V tile. One DMA instruction = 4 rows. Wave w issues
    //      pieces 2w, 2w+1 (rows 8w .. 8w+7) of K and of V. lane -> (row = 4 piece + (lane>>4), stored slot =
    //      lane & 15); the stored slot holds logical slot  s ^ (row & 15)  (K)  /  s ^ ((row & 3) << 2)  (V).
    const int nkv_blk = min(lim_blk / KT + 1, (T_ + KT - 1) / KT);                     // keys <= the last row's upper edge
    int drow[2], dks[2], dvs[2];
    unsigned koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + (lane >> 4);
        drow[i] = row;
        dks[i] = slot_in((lane & 15) ^ (row & 15), p.D) * 16;
        dvs[i] = slot_in((lane & 15) ^ ((row & 3) << 2), p.D) * 16;
        koff[i] = (unsigned)((int64_t)row * p.k_st * 2 + dks[i]);
        voff[i] = (unsigned)((int64_t)row * p.v_st * 2 + dvs[i]);
    }
    const int kvr = p.kvm == 1 ? kvh : kvh / p.kvm;                                     // the K / V head behind a virtual one
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvr * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvr * p.v_sh;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 2048;
    auto issue = [&](int t, int stage) {
        const int k0 = t * KT;
        const unsigned d = dst_w + stage * STAGE_B;
        if (k0 + KT <= T_) {
            dma16x2(kbase + (int64_t)k0 * p.k_st, koff[0], koff[1], d, d + 1024);
            dma16x2(vbase + (int64_t)k0 * p.v_st, voff[0], voff[1], d + TILE_B, d + TILE_B + 1024);
        } else {
            // ragged last tile: rows past the end re-read the last key (they are masked)
            unsigned ko[2], vo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = min(drow[i], T_ - 1 - k0);
                ko[i] = (unsigned)((int64_t)r * p.k_st * 2 + dks[i]);
                vo[i] = (unsigned)((int64_t)r * p.v_st * 2 + dvs[i]);
            }
            dma16x2(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], d, d + 1024);
            dma16x2(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], d + TILE_B, d + TILE_B + 1024);
        }
    };

    // ---- per-lane LDS addresses
    // K A-operand (lane -> key l31 (+32 kt), 16 B at slot 2 ks + lh): (l31*256 + x'*16) ^ (ks*32)
    const int kx = l31 & 15;
    const int k_lane = l31 * 256 + (((kx & 14) | (lh ^ (kx & 1))) << 4);
    // V^T A-operand via ds_read_b64_tr_b16. 16-lane group g = lane>>4 -> (d half = g&1, key half = g>>1 = lh);
    // lane s = lane&15 supplies row (s>>2) of the 4-row block, 8 B at columns 4 (s&3).
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int v_lane = (4 * lh + (sg >> 2)) * 256 + ((((sg >> 2) << 2) | (gh << 1) | ((sg >> 1) & 1)) << 4) + (sg & 1) * 8;

    f32x16_t o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;      // running max (log2 domain, both lane halves agree) / partial sum

#ifdef UAMD_ATTN_TRACE
    unsigned ats[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ats[i] = 0;
#endif
    // ---- prologue
    const int nt = nkv_blk - t_first;
    issue(t_first, 0);
    if (nt > 1) issue(t_first + 1, 1);

    const int last_tile_wave = lim_w1 / KT;                          // tiles beyond are fully masked for this wave
    const int first_tile_wave = lo_w0 / KT;                          // ... and tiles before
    // One tile step; MASKED is compile-time and the tile range is split by hand (see attn_bwd_dq_kernel).
    // DC (template) = head-dim class, 64 / 96 / 128 columns: the k-steps of S^T and the d-tiles of O^T past it are not computed (the
    // Q fragments there are zero and the rows are never stored anyway: same results, half the MFMAs and LDS reads at head_dim 64)
    auto step = [&](int ti, auto mode_c, bool rt_mask) {       // mode 0: no mask, 1: mask, 2: mask iff rt_mask
        constexpr int MODE = decltype(mode_c)::value;
        constexpr int NKS = DC / 16, NDT = DC / 32;
        const int t = t_first + ti;
        ASTAMP(ti, 0);
        if (ti + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ASTAMP(ti, 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ASTAMP(ti, 2);
        if (ti + 2 < nt) issue(t + 2, (ti + 2) % NST);              // its stage was last read before this barrier
        ASTAMP(ti, 3);
        if (t > last_tile_wave || t < first_tile_wave) return;       // wave-uniform: nothing to add
        const unsigned char* sk = smem + (ti % NST) * STAGE_B;
        const unsigned char* sv = sk + TILE_B;

        // ---- S^T[key][q] = K Q^T : 2 key tiles x 8 k-steps
        f32x16_t st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                union { uint4 r; frag_t f; } u;
                u.r = *reinterpret_cast<const uint4*>(sk + kt * 32 * 256 + (k_lane ^ (ks * 32)));
                st[kt] = MfmaA<T>::run(u.f, qf[ks], st[kt]);
            }
        }
        ASTAMP(ti, 4);
        // ---- (plain-causal build, which has the registers for it) the V^T operands of the first two 16-key steps
        //      are fetched NOW: the transposing reads do not depend on P and the softmax below is pure VALU --
        //      otherwise each of the 16 PV MFMAs waits for its own two ds_read_b64_tr_b16 (tools/attn_trace.py:
        //      2,500 cycles for 16 MFMAs; 1,600 with the operands two steps ahead). The tile period only moves from
        //      6,200 to 5,800 cycles: with 8 waves per CU the LDS read stream itself (256 KB per tile step, half of
        //      it 8-byte transposing reads that need >= 4 waves per SIMD for full rate) is the next limit.
        constexpr bool PREFETCH = !BAND;
        auto load_v = [&](int u, frag_t* dst) {                       // u = 2 kt + c: keys 16 u .. 16 u + 15
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int a0 = (u * 16) * 256 + (v_lane ^ (dt << 6));
                union { s16x4_t h[2]; frag_t f; } va;
                va.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0));
                va.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0 + 8 * 256));
                dst[dt] = va.f;
            }
        };
        frag_t va0[4], va1[4], va2[4];
        if (PREFETCH) {
            load_v(0, va0);
            load_v(1, va1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- online softmax, log2 domain. lane: q = q_pos; register r of tile kt: key below
        // MFMA time and (non-transcendental) VALU time ADD UP on a gfx950 SIMD (profiles/r02_mfma_valu_overlap.txt), so the
        // softmax is written for instruction count: the scale rides in the exponent's fma (max(s) c = max(s c), c > 0),
        // and O / l are rescaled only when some row's max rose (wave-uniform test; after the first tiles it rarely does)
        const int k0 = t * KT;
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (MODE == 1 || (MODE == 2 && rt_mask)) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > lim_q || key >= T_ || key < lo_q) st[kt][r] = -INFINITY;
                }
                mt = fmaxf(mt, st[kt][r]);
            }
        mt = max_across_halves(mt) * p.scale_log2;
        if (__builtin_amdgcn_ballot_w64(mt > m_run) != 0) {
            const float m_new = fmaxf(m_run, mt);
            // a row whose band starts after this tile has seen only masked keys so far: keep the exponent finite
            const float alpha = __builtin_amdgcn_exp2f(m_run - (m_new == -INFINITY ? 0.f : m_new));   // first tile: exp2(-inf) = 0
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDT; ++i) o_acc[i] *= alpha;
        }
        const float m_ref = m_run == -INFINITY ? 0.f : m_run;
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], p.scale_log2, -m_ref));
                st[kt][r] = e;
                ls += e;
            }
        l_run += ls;

        ASTAMP(ti, 5);
        // ---- O^T[d][q] += V^T P^T : per 16-key step u = 2 kt + c; lane half lh contracts keys
        //      16 u + {4 lh .. 4 lh + 3, 8 + 4 lh .. 8 + 4 lh + 3} = registers 8c .. 8c+7 of st[kt]
        auto pv = [&](int u, const frag_t* vsrc) {
            const int kt = u >> 1, c = u & 1;
            union { uint32_t w[4]; frag_t f; } pb;
#pragma unroll
            for (int j = 0; j < 4; ++j) pb.w[j] = pack_pair2<T>(st[kt][8 * c + 2 * j], st[kt][8 * c + 2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o_acc[dt] = MfmaA<T>::run(vsrc[dt], pb.f, o_acc[dt]);
        };
        if (PREFETCH) {                                               // operand fetch two steps ahead, 3 buffers
            load_v(2, va2);
            __builtin_amdgcn_sched_barrier(0);
            pv(0, va0);
            __builtin_amdgcn_sched_barrier(0);
            load_v(3, va0);
            __builtin_amdgcn_sched_barrier(0);
            pv(1, va1);
            pv(2, va2);
            pv(3, va0);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_v(u, va0);
                pv(u, va0);
            }
        }
        ASTAMP(ti, 6);
    };
    {
        const int t_pre_end = min(nkv_blk, (lo_w1 + KT - 1) / KT);       // tiles that start below the band edge
        const int t_diag = (lim_w0 + 1) / KT, t_rag = (T_ % KT) ? T_ / KT : nkv_blk;
        const int t_suf = max(t_pre_end, min(min(t_diag, t_rag), nkv_blk));
        if constexpr (BAND) {
            int ti = 0;
            for (; t_first + ti < t_pre_end; ++ti) step(ti, std::integral_constant<int, 1>{}, true);
            for (; t_first + ti < t_suf; ++ti) step(ti, std::integral_constant<int, 0>{}, false);
            for (; ti < nt; ++ti) step(ti, std::integral_constant<int, 1>{}, true);
        } else {
            // plain causal: ONE loop with the (monotone) mask test around the masking statement only -- hipcc
            // splits the range itself and needs far fewer registers than with the hand-split loops
            for (int ti = 0; ti < nt; ++ti) step(ti, std::integral_constant<int, 2>{}, ti >= t_suf);
        }
    }

#ifdef UAMD_ATTN_TRACE
    if (g_attn_trace && lane == 0 && blockIdx.x < 256) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g_attn_trace[(blockIdx.x * 8 + wave) * 16 + i] = ats[i];
    }
#endif
    // ---- epilogue: O = O^T / l, LSE = ln2 * (m + log2 l)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    {
        T* op = (T*)p.O + b * p.o_sb + (int64_t)q_ld * p.o_st + (int64_t)head * p.o_sh;
        if (p.D == AD) store_rows_x4<T>(op, o_acc, inv, lh, q_pos < T_);
        else store_rows_x4<T>(op, o_acc, inv, lh, q_pos < T_, p.D);
        if (lh == 0 && q_pos < T_) p.LSE[((int64_t)b * p.Hq + head) * p.lse_st + q_pos] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Forward, PERSISTENT PING-PONG kernel (round 4): plain causal batches with >= 2 work items per CU. Same tiling, LDS image,
// swizzles and DMA ring as attn_fwd_kernel; what changes is WHEN a wave does what, and that a workgroup does not end after one
// (q tile, batch, KV head) item.
//
// Ping-pong. In attn_fwd_kernel all 8 waves run [K reads + S MFMAs | softmax | V reads + PV MFMAs] behind one barrier per tile:
// the two waves of a SIMD want the matrix pipe in the same phases and leave it idle in the same phases, and every MFMA waits
// for its own LDS read. Here a tile is four phases separated by block barriers,
//     P1  K rows of the tile LDS -> 64 registers (16 ds_read_b128)
//     P2  S^T = K Q^T: 16 MFMAs on registers only; in their shadow the LDS-DMA of the tile two ahead in the stream
//     P3  V^T fragments LDS -> the SAME 64 registers (32 ds_read_b64_tr_b16); mask, row max (a tree), rescale test, P^T step 0
//     P4  O^T += V^T P^T: 16 MFMAs on registers only; the exponentials of step u + 1 in the shadow of step u's four MFMAs
// and waves 4-7 (the second wave of every SIMD: a workgroup's waves go to the SIMDs in cyclic order) run ONE PHASE BEHIND
// waves 0-3: a matrix phase of one wave sits beside a load phase of its SIMD partner (MI355X_MICROARCH.md "Two waves per
// SIMD": matrix beside memory, never matrix beside matrix). Ring safety under the skew: the stage of stream tile s is last read
// in the trailing group's P3(s), which ends at the barrier before the leading group's P1(s + 1); the first DMA into that stage
// (tile s + 3) is issued in P2(s + 1); every wave waits for ITS pieces of tile s + 1 at the end of its P3(s), one barrier before
// anybody reads them. Measured phase by phase with s_memtime (profiles/r04_attn_fwd_pp_phase_and_block_timeline.txt): P1 480,
// P2 840-980, P3 1,440, P4 950-1,020 cycles -- a wave issues ~1 instruction per 4-5 cycles whatever its partner does, so the
// 84 softmax instructions moved from P3 into P4's MFMA shadow lengthened P4 by what they shortened P3; the tile step is 4,500
// cycles in this kernel, in attn_fwd_kernel and in the 4-wave x 64-row kernel of round 2 alike
// (profiles/r04_attn_ab_three_forward_kernels.jsonl), 3,750 with the K/V DMA compiled out, 4,800 with every tile read from the
// same L2-resident addresses. The ping-pong alone (one block per item: git 4501bb3:tools/experiments/attention_removed_r04.hip) was 4 %
// SLOWER than the lockstep kernel: the same tile step plus two more barriers of pipeline fill per block.
//
// Persistence. At 4 x 2048 tokens a block's tile loop is 16.5 tiles, and every block pays ~28,000 cycles that do not depend on
// its length (two-shape fit in the same profile; block timeline: 9,100 waiting for its Q fragments + first two K/V tiles, ~1,000
// between one workgroup's end and the next one's entry, 4,300 of epilogue, the pipeline fill, and its share of the kernel's
// tail -- the last blocks of the heaviest-first order still cost their fixed part) -- a quarter of the kernel. Here
//   * the K/V ring runs THROUGH the seams: the DMA slot of a tile fetches "the tile two ahead in the stream", which during an
//     item's last two tiles is the next item's first two tiles;
//   * the next item's Q rows arrive by LDS-DMA as whole 256-byte rows (the per-lane 16-byte gathers of the one-shot kernels
//     touch every 128-byte line four times) into a 64 KiB staging area beside the ring -- each wave its own 8 KiB, written and
//     read by that wave only -- during the current item's first tile; the fragments are read from there at the seam;
//   * items are dealt to the workgroups in snake order over the heaviest-first list, so every workgroup gets the same number
//     of tile steps (plain causal: exactly).
// Result: +7 % over the one-shot ping-pong, +2-3 % over attn_fwd_kernel (profiles/r04_attn_ab_persistent.jsonl); what is left per
// item is the epilogue (store-issue-bound) and the first tile's fill. LDS: 96 KiB ring + 64 KiB Q staging = 160 KiB.
constexpr int QSTAGE_OFF = NST * STAGE_B;                 // 96 KiB
constexpr int ATTN_PS_LDS = QSTAGE_OFF + 8 * 8192;        // 160 KiB

template <typename T, bool BAND, bool DYN>
__global__ void __launch_bounds__(512, 2) attn_fwd_ps_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 32 * p.nsub;
    const int npairs = p.Hk * p.B;
    const int nitems = p.nqt * npairs, nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    // item k of this workgroup: snake over the heaviest-first list
    auto item_index = [&](int k) { return k * nwg + ((k & 1) ? nwg - 1 - wg : wg); };
    struct Item { int qtile, kvh, kvr, b, t_first, nt; };            // kvr: the K / V head behind the (virtual) head kvh
    // first key tile of item idx (band: the lower edge of the item's first query)
    auto first_tile = [&](int idx) {
        if (!BAND) return 0;
        const int qtile = p.nqt - 1 - idx / npairs;
        return p.lo[(int64_t)((idx % npairs) / p.Hk) * T_ + min(qtile * QT, T_ - 1)] / KT;
    };
    auto decode_at = [&](int idx, int t_first) {
        Item it;
        it.qtile = p.nqt - 1 - idx / npairs;
        const int pair_ = idx % npairs;
        it.kvh = pair_ % p.Hk;
        it.kvr = p.kvm == 1 ? it.kvh : it.kvh / p.kvm;
        it.b = pair_ / p.Hk;
        it.t_first = t_first;
        it.nt = min((it.qtile * QT + QT + KT - 1) / KT, (T_ + KT - 1) / KT) - it.t_first;
        return it;
    };
    auto decode = [&](int idx) { return decode_at(idx, first_tile(idx)); };
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 2048;
    const unsigned qdst_w = lds_base + QSTAGE_OFF + wave * 8192;
    // one K/V tile of item `it` into ring stage `stage` (this wave's 4 pieces)
    // DMA plan: a stage = K tile (64 rows x 256 B) then V tile; one instruction = 4 rows; wave w issues pieces 2w, 2w + 1 (rows
    // 8w .. 8w + 7) of K and of V; lane -> (row = 4 piece + (lane >> 4), stored slot = lane & 15), the stored slot holds logical
    // slot s ^ (row & 15) (K) / s ^ ((row & 3) << 2) (V). The per-lane source offsets are RECOMPUTED at every issue from an
    // opaque copy of the lane id (a dozen VALU instructions per tile): kept in registers across the tile loop they are the first
    // thing hipcc spills, and a scratch reload in front of the DMA drains vmcnt (= the ring)
    auto issue_tile = [&](const Item& it, int t, int stage) {
        const T* kbase = (const T*)p.K + it.b * p.k_sb + (int64_t)it.kvr * p.k_sh;
        const T* vbase = (const T*)p.V + it.b * p.v_sb + (int64_t)it.kvr * p.v_sh;
        const int k0 = t * KT;
        const unsigned d = dst_w + stage * STAGE_B;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int rmax = T_ - 1 - k0;
        unsigned ko[2], vo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + (ln >> 4);
            const int r = min(row, rmax);
            ko[i] = (unsigned)(r * (int)p.k_st * 2 + ((ln & 15) ^ (row & 15)) * 16);
            vo[i] = (unsigned)(r * (int)p.v_st * 2 + ((ln & 15) ^ ((row & 3) << 2)) * 16);
        }
        dma16x2(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], d, d + 1024);
        dma16x2(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], d + TILE_B, d + TILE_B + 1024);
    };
    // this wave's 32 Q rows of item `it` (8 pieces of 4 rows x 256 B) into its staging area, K-row swizzle on the source
    auto issue_q = [&](const Item& it) {
        const int head = it.kvh * G + (wave % G);
        const int qs = it.qtile * QT + (wave / G) * 32;
        const T* qbase = (const T*)p.Q + it.b * p.q_sb + (int64_t)head * p.q_sh;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        unsigned qo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 4 + (ln >> 4);
            const int r = min(qs + row, T_ - 1);                       // rows past the end re-read the last one (never stored)
            qo[i] = (unsigned)r * (unsigned)(p.q_st * 2) + (unsigned)(((ln & 15) ^ (row & 15)) * 16);    // (host: T * q_st < 2^31)
        }
        dma16x4g(qbase, qo[0], qo[1], qo[2], qo[3], qdst_w);
        dma16x4g(qbase, qo[4], qo[5], qo[6], qo[7], qdst_w + 4096);
    };
    // The common case -- tile ti + 2 of the CURRENT item, all 64 keys inside the sequence -- takes a short issue: the four
    // per-lane source offsets are kernel constants (no row clamp), the tile's K / V addresses are running pointers advanced by a
    // constant, the ring stage a counter. The general issue above costs ~26 VALU + ~40 SALU instructions per tile step (64-bit
    // base products, a division by 3 through mul_hi, the clamp), this one ~12; the kernel is bound by its instruction count
    // (DESIGN 5b), and at 237 VGPRs the four offsets no longer spill.
    unsigned kof[2], vof[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + (lane >> 4);
        kof[i] = (unsigned)(row * (int)p.k_st * 2 + ((lane & 15) ^ (row & 15)) * 16);
        vof[i] = (unsigned)(row * (int)p.v_st * 2 + ((lane & 15) ^ ((row & 3) << 2)) * 16);
    }
    const int64_t kstep = (int64_t)KT * p.k_st, vstep = (int64_t)KT * p.v_st;       // elements per tile
    auto issue_fast = [&](const T* kq, const T* vq, int stage) {
        const unsigned d = dst_w + stage * STAGE_B;
        dma16x2(kq, kof[0], kof[1], d, d + 1024);
        dma16x2(vq, vof[0], vof[1], d + TILE_B, d + TILE_B + 1024);
    };
    const int kx = l31 & 15;
    const int k_lane = l31 * 256 + (((kx & 14) | (lh ^ (kx & 1))) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int v_lane = (4 * lh + (sg >> 2)) * 256 + ((((sg >> 2) << 2) | (gh << 1) | ((sg >> 1) & 1)) << 4) + (sg & 1) * 8;
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    frag_t qf[8];
    auto read_q = [&]() {                 // Q^T operand fragments from the wave's staging area (same lane map as the K rows)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            union { uint4 r; frag_t f; } u;
            u.r = *reinterpret_cast<const uint4*>(smem + QSTAGE_OFF + wave * 8192 + (k_lane ^ (ks * 32)));
            qf[ks] = u.f;
        }
        wait_lgkm0();
    };

    // DYN (packed / windowed batches: items of one q tile differ in length by what the band cuts off, the static deal is off by
    // up to 2x): item 0 of a workgroup is its index, every further item is CLAIMED from a counter in heaviest-first order
    // (later q tiles cannot be long: a late claim is a short item). The ring and the Q staging need the next item while the
    // current one runs, so a workgroup always holds one claimed item ahead of the one it works on:
    //   * wave 0 claims (relaxed agent-scope fetch-add; LLVM's atomic optimizer makes it one lane's add + a v_readfirstlane, so
    //     the wait for the result sits right behind the atomic: ~ one L2 round trip, in one wave, once per item) at the START
    //     OF AN ITEM'S EPILOGUE -- two phases behind the last LDS-DMA issue, one before the next: the compiler's vmcnt for the
    //     result counts only the memory operations it knows of, and between P3 and the next P2 there is none it does not --,
    //     reads the claimed item's band edge behind the epilogue's stores, and
    //   * publishes (item, first key tile) in P1 of the next item's first tile through 8 bytes of the ring stage that tile
    //     step's DMA will fill: free since the trailing group's P3 of the previous tile, written by wave 7 only in ITS P2 --
    //     the leading waves read it at the top of their P2, the trailing waves (one phase behind) in their P1, all between the
    //     same two barriers. All 160 KiB of LDS are taken; the hand-off borrows.
    // Exactly one claim per workgroup fails (>= nitems), then it claims no more. The last workgroup to finish zeroes the pair.
    auto mailbox = [&](int stage) { return smem + stage * STAGE_B + 7 * 2048; };
    int pend_v = 0;                                      // wave 0, lane 0: the fetch-add's result, not yet waited for
    int pend = 0x7fffffff, pend_tf = 0;                  // wave 0: the claimed item and its first key tile, to be published
    auto claim = [&]() {
        if (lane == 0) pend_v = __hip_atomic_fetch_add(p.ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto claimed = [&]() {
        pend = nwg + __builtin_amdgcn_readfirstlane(pend_v);
        pend_tf = pend < nitems ? first_tile(pend) : 0;
    };
    if (DYN && wave == 0) claim();
    if (item_index(0) >= nitems) return;                 // (whole workgroup: no barrier was executed yet; never under DYN: nwg <= nitems)
    Item cur = decode(item_index(0));
    // ---- prologue of the first item: Q rows, tiles 0 and 1
    issue_q(cur);
    issue_tile(cur, cur.t_first, 0);
    if (cur.nt > 1) issue_tile(cur, cur.t_first + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DYN && wave == 0) claimed();     // (the fetch-add was issued ahead of the prologue's DMA: it has returned)
    bar();
    if (wave >= 4) bar();                // the trailing group drops one phase behind
    int s_ = 0;                          // stream index of the current item's first tile (ring stage = stream index % 3)
    int stg = 0;                         // ring stage of the CURRENT tile = (s_ + ti) % 3, counted instead of divided
    int kitem = 0;
    int pre = 2;                         // leading tiles of the current item that are already fetched (the prologue: both)
    frag_t kv[16];
    f32x16_t st[2];
    frag_t pb[4];
    f32x16_t o_acc[4];
    for (;;) {
        // the item after this one: by the static deal, or (DYN) taken from the mailbox in this item's first tile
        int nxt_idx = DYN ? 0x7fffffff : item_index(kitem + 1);
        bool has_next = nxt_idx < nitems;
        Item nxt = cur;
        if (!DYN && has_next) nxt = decode(nxt_idx);
        auto take_next = [&](int stage) {
            const unsigned long long m = *reinterpret_cast<const volatile unsigned long long*>(mailbox(stage));
            nxt_idx = __builtin_amdgcn_readfirstlane((int)(unsigned)m);
            has_next = nxt_idx < nitems;
            if (has_next) nxt = decode_at(nxt_idx, __builtin_amdgcn_readfirstlane((int)(unsigned)(m >> 32)));
        };
        const int head = cur.kvh * G + (wave % G);
        const int qs = cur.qtile * QT + (wave / G) * 32;
        const int q_pos = qs + l31;
        const int q_ld = q_pos < T_ ? q_pos : T_ - 1;
        const int lo_q = BAND ? p.lo[(int64_t)cur.b * T_ + q_ld] : 0;
        const int lo_w0 = BAND ? __builtin_amdgcn_readfirstlane(lo_q) : 0, lo_w1 = BAND ? __builtin_amdgcn_readlane(lo_q, 31) : 0;
        const int nkv_blk = cur.t_first + cur.nt;
        const int last_tile_wave = min(qs + 31, T_ - 1) / KT;
        const int first_tile_wave = lo_w0 / KT;
        const int t_pre_end = min(nkv_blk, (lo_w1 + KT - 1) / KT);
        const int t_diag = (qs + 1) / KT, t_rag = (T_ % KT) ? T_ / KT : nkv_blk;
        const int t_suf = max(t_pre_end, min(min(t_diag, t_rag), nkv_blk));
        const int nt = cur.nt;
        // K / V rows of tile ti + 2 (running: + one tile per step)
        const T* kq = (const T*)p.K + cur.b * p.k_sb + (int64_t)cur.kvr * p.k_sh + (int64_t)(cur.t_first + 2) * kstep;
        const T* vq = (const T*)p.V + cur.b * p.v_sb + (int64_t)cur.kvr * p.v_sh + (int64_t)(cur.t_first + 2) * vstep;
        read_q();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        auto softmax_piece = [&](int u, float m_ref, float& ls0, float& ls1) {
            const int kt = u >> 1, c = u & 1;
            union { uint32_t w[4]; frag_t f; } w_;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][8 * c + 2 * jj], p.scale_log2, -m_ref));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][8 * c + 2 * jj + 1], p.scale_log2, -m_ref));
                ls0 += e0;
                ls1 += e1;
                w_.w[jj] = pack_pair2<T>(e0, e1);
            }
            pb[u] = w_.f;
        };
        for (int ti = 0; ti < nt; ++ti) {
            const int t = cur.t_first + ti;
            const int sx = s_ + ti;                                              // stream index
            const bool live = !(t > last_tile_wave || t < first_tile_wave);     // wave-uniform
            const unsigned char* sk = smem + stg * STAGE_B;
            const unsigned char* sv = sk + TILE_B;
            const int stg2 = stg == 0 ? 2 : stg - 1;                            // (sx + 2) % 3
            // ---------------- P1 (load): K rows -> 64 registers
            if (DYN && ti == 0) {
                if (wave == 0) {
                    if (lane == 0)
                        *reinterpret_cast<volatile unsigned long long*>(mailbox(stg2)) =
                            (unsigned long long)(unsigned)pend | ((unsigned long long)(unsigned)pend_tf << 32);
                    wait_lgkm0();
                } else if (wave >= 4) {
                    take_next(stg2);                                            // (published one barrier ago)
                }
            }
            if (live) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        union { uint4 r; frag_t f; } u;
                        u.r = *reinterpret_cast<const uint4*>(sk + kt * 32 * 256 + (k_lane ^ (ks * 32)));
                        kv[kt * 8 + ks] = u.f;
                    }
                wait_lgkm0();
            }
            bar();
            // ---------------- P2 (matrix): S^T = K Q^T; in its shadow the DMA of stream tile sx + 2 (this item's tile ti + 2, or
            //                  the NEXT item's tile ti + 2 - nt) and, once per item, the next item's Q rows
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[0][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) st[0] = MfmaA<T>::run(kv[ks], qf[ks], st[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // DMA order inside this slot: next item's Q rows, then a catch-up tile, then the tile two ahead -- the wait at the end
            // of P3 keeps only the LAST group in flight
            if (DYN && ti == 0 && wave < 4) take_next(stg2);
            bool issued = false;
            if (ti == 0 && has_next) issue_q(nxt);                 // (this item's read_q is long done)
            if (ti == 0 && pre == 1 && nt > 1) issue_tile(cur, t + 1, (sx + 1) % NST);      // predecessor had a single tile
            if (ti + 2 < nt) {
                if ((t + 3) * KT <= T_) issue_fast(kq, vq, stg2);
                else issue_tile(cur, t + 2, stg2);                    // ragged last tile of the sequence: rows clamped
                kq += kstep;
                vq += vstep;
                issued = true;
            } else if (has_next) {
                // the next item's tile j = stream tile s_ + nt + j. A single-tile item has only this one slot: it fetches the
                // next item's tile 0 (needed one tile from now), and that item catches up on its tile 1 itself
                const int j = nt == 1 ? 0 : ti + 2 - nt;
                if (j < nxt.nt) {
                    issue_tile(nxt, nxt.t_first + j, (s_ + nt + j) % NST);
                    issued = true;
                }
            }
            if (live) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 4; ks < 8; ++ks) st[0] = MfmaA<T>::run(kv[ks], qf[ks], st[0]);
                // (zeroed HERE, in the block of its first MFMA: the C operand becomes the inline constant 0 -- set in the block
                // above, across the DMA code, hipcc built the 16 zeros through 16 s_mov + 8 v_mov_b64 per tile step)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[1][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) st[1] = MfmaA<T>::run(kv[8 + ks], qf[ks], st[1]);
            }
            bar();
            // ---------------- P3 (load + row statistics)
            float m_ref = 0.f, ls0 = 0.f, ls1 = 0.f;
            if (live) {
                const int k0 = t * KT;
                const bool need_mask = BAND ? (t < t_pre_end || t >= t_suf) : (t >= t_suf);
                if (need_mask) {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            if (key > q_pos || key >= T_ || key < lo_q) st[kt][r] = -INFINITY;
                        }
                }
                float m8[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int a0 = (u * 16) * 256 + (v_lane ^ (dt << 6));
                        union { s16x4_t h[2]; frag_t f; } va;
                        va.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0));
                        va.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0 + 8 * 256));
                        kv[u * 4 + dt] = va.f;
                    }
                    const int kt = u >> 1, c = u & 1;
#ifdef UAMD_ATTN_MAX_TREE2
                    m8[2 * u] = fmaxf(fmaxf(st[kt][8 * c], st[kt][8 * c + 1]), fmaxf(st[kt][8 * c + 2], st[kt][8 * c + 3]));
                    m8[2 * u + 1] = fmaxf(fmaxf(st[kt][8 * c + 4], st[kt][8 * c + 5]), fmaxf(st[kt][8 * c + 6], st[kt][8 * c + 7]));
#else
                    // four independent chains of v_max3_f32 (a chain link takes in TWO new scores): 18 instructions for the 32
                    // scores of a lane instead of the pairwise tree's 31 -- the kernel is bound by instruction count
#define SP_(i) st[kt][8 * c + (i)]
                    if (u == 0) { m8[0] = max3f(SP_(0), SP_(1), SP_(2)); m8[1] = max3f(SP_(3), SP_(4), SP_(5)); m8[2] = SP_(6); m8[3] = SP_(7); }
                    else if (u == 1) { m8[2] = max3f(m8[2], SP_(0), SP_(1)); m8[3] = max3f(m8[3], SP_(2), SP_(3));
                                       m8[0] = max3f(m8[0], SP_(4), SP_(5)); m8[1] = max3f(m8[1], SP_(6), SP_(7)); }
                    else { m8[0] = max3f(m8[0], SP_(0), SP_(1)); m8[1] = max3f(m8[1], SP_(2), SP_(3));
                           m8[2] = max3f(m8[2], SP_(4), SP_(5)); m8[3] = max3f(m8[3], SP_(6), SP_(7)); }
#undef SP_
#endif
                }
#ifdef UAMD_ATTN_MAX_TREE2
                float mt = fmaxf(fmaxf(fmaxf(m8[0], m8[1]), fmaxf(m8[2], m8[3])), fmaxf(fmaxf(m8[4], m8[5]), fmaxf(m8[6], m8[7])));
#else
                float mt = fmaxf(max3f(m8[0], m8[1], m8[2]), m8[3]);
#endif
                mt = max_across_halves(mt) * p.scale_log2;
                if (__builtin_amdgcn_ballot_w64(mt > m_run) != 0) {
                    const float m_new = fmaxf(m_run, mt);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - (m_new == -INFINITY ? 0.f : m_new));
                    m_run = m_new;
                    l_run *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o_acc[i] *= alpha;
                }
                m_ref = m_run == -INFINITY ? 0.f : m_run;
                softmax_piece(0, m_ref, ls0, ls1);
                __builtin_amdgcn_sched_barrier(0);
                wait_lgkm0();
            }
            // stream tile sx + 1 has landed (this wave's pieces) one barrier before anybody reads it. It was issued one tile ago --
            // or in THIS tile's P2 when the item has a single tile (then everything is waited for)
            if (issued && nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
            // ---------------- P4 (matrix): O^T += V^T P^T
            if (live) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o_acc[dt] = MfmaA<T>::run(kv[u * 4 + dt], pb[u], o_acc[dt]);
                    if (u < 3) softmax_piece(u + 1, m_ref, ls0, ls1);
#pragma unroll
                    for (int g_ = 0; g_ < 4; ++g_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                    }
                }
                l_run += ls0 + ls1;
            }
            bar();
            stg = stg == NST - 1 ? 0 : stg + 1;
        }
        // ---- epilogue of the item: O = O^T / l, LSE
        if (DYN && wave == 0) {
            if (has_next) { claim(); claimed(); }                               // the item after the next one (its band edge arrives
            else pend = 0x7fffffff;                                             // behind the stores below)
        }
        {
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            const float inv = 1.0f / l_tot;
            int qr = q_ld;
            asm volatile("" : "+v"(qr));
            T* op = (T*)p.O + cur.b * p.o_sb + (int64_t)qr * p.o_st + (int64_t)head * p.o_sh;
            store_rows_x4<T>(op, o_acc, inv, lh, q_pos < T_);
            if (lh == 0 && q_pos < T_) p.LSE[((int64_t)cur.b * p.Hq + head) * p.lse_st + qr] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
        }
        if (!has_next) break;
        pre = nt == 1 ? 1 : 2;
        s_ += nt;
        cur = nxt;
        ++kitem;
    }
    if (wave < 4) bar();                 // the leading group's extra barrier = the trailing group's last phase
    if (DYN && wave == 0 && lane == 0) {
        // every workgroup's claims precede its arrival here (same wave, results consumed): the last one in resets the pair
        if (__hip_atomic_fetch_add(p.ctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            __hip_atomic_store(p.ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.ctr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Backward, part 1: dQ (Q-stationary, same tiling as the forward) + Delta[q] = sum_d dO[q][d] O[q][d].
//   S^T = K Q^T, P^T = exp2(S^T c - LSE2[q]), dP^T = V dO^T, dS^T = P^T (dP^T - Delta[q]) * scale,
//   dQ^T[d][q] += K^T[d][key] dS^T[key][q]
// K is read two ways from the same LDS tile: by rows (A operand of S^T, ds_read_b128) and transposed (A operand
// of dQ^T, ds_read_b64_tr_b16). The swizzle  slot ^= ((row & 3) << 2) | ((row >> 2) & 3)  is conflict-free for
// both: injective over row & 15 (row reads), and the four rows of a transposing read land in four different
// 64-byte windows.
struct AttnBwdArgs {
    const void* Q; const void* K; const void* V; const void* O; const void* dO; const float* LSE;
    void* dQ; void* dK; void* dV; float* Delta;
    const int* lo; const int* hi;        // band: query q attends keys lo[q] <= key <= q  <=>  q <= hi[key]
    int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh, do_sb, do_st, do_sh;
    int64_t dq_sb, dq_st, dq_sh, dk_sb, dk_st, dk_sh, dv_sb, dv_st, dv_sh;
    int B, T, Hq, Hk, G, nsub, lse_st, nqt;
    int D;                               // head dim (multiple of 8, <= 128; AttnArgs::D): operands past D are zero REGISTERS on one side
                                         // of every product (Q, dO in the dQ kernel; K in the dK / dV kernel, whose V tile is zeroed
                                         // past D in LDS), rows past D of dQ^T / dK^T / dV^T are never stored
    int kvm;                             // attn_bwd_dq_kernel: Hk counts virtual KV heads (AttnArgs::kvm); the dK / dV kernel takes the
                                         // real heads with G = Hq / Hk of any size 1 .. 8 (passes of 4, 2 and 1 query heads)
    float scale, scale_log2;
    int noncausal;                       // 1: bidirectional inside documents -- query q and key k attend iff lo[q] <= k <= hi[q]
                                         // (documents are intervals, so equivalently lo[k] <= q <= hi[k]); needs lo AND hi
    int no_asm;                          // attn_bwd_dkdv4_kernel: 1 = no step takes the generated loop (UAMD_TUNE_ATTN_VAR bit 2)
};

__device__ __forceinline__ int swz_c(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <typename T, bool BAND, int DC>
__global__ void __launch_bounds__(512, 2) attn_bwd_dq_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 32 * p.nsub;
    const int npairs = p.Hk * p.B;
    int rank_, pair_;
    block_to_work((int)blockIdx.x, npairs, rank_, pair_);
    const int qtile = p.nqt - 1 - rank_;                              // heaviest q tiles first
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int head = kvh * G + (wave % G);
    const int qs = qtile * QT + (wave / G) * 32;
    const int q_pos = qs + l31;
    const int q_ld = q_pos < T_ ? q_pos : T_ - 1;

    frag_t qf[8], dof[8];
    float delta = 0.f;
    {
        const T* qp = (const T*)p.Q + b * p.q_sb + (int64_t)q_ld * p.q_st + (int64_t)head * p.q_sh + lh * 8;
        const T* dp_ = (const T*)p.dO + b * p.do_sb + (int64_t)q_ld * p.do_st + (int64_t)head * p.do_sh + lh * 8;
        const T* op = (const T*)p.O + b * p.o_sb + (int64_t)q_ld * p.o_st + (int64_t)head * p.o_sh + lh * 8;
        // MASKED = head dims below 128: zero registers past D. The loads stay unconditional (a lane past D re-reads the row's
        // first 16 bytes) and the zeroing is an AND, see attn_fwd_kernel; head_dim 128 (a wave-uniform test) keeps the plain loads.
        auto load_rows = [&](auto masked_c) __attribute__((always_inline)) {
            constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                union { uint4 r; frag_t f; T e[8]; } u, d, o;
                const bool in = !MASKED || ks * 16 + lh * 8 < p.D;
                const int co = in ? ks * 16 : -lh * 8;
                u.r = *reinterpret_cast<const uint4*>(qp + co);
                d.r = *reinterpret_cast<const uint4*>(dp_ + co);
                o.r = *reinterpret_cast<const uint4*>(op + co);
                if constexpr (MASKED) {
                    const unsigned keep = in ? 0xffffffffu : 0u;
                    u.r.x &= keep; u.r.y &= keep; u.r.z &= keep; u.r.w &= keep;
                    d.r.x &= keep; d.r.y &= keep; d.r.z &= keep; d.r.w &= keep;
                }
                qf[ks] = u.f;
                dof[ks] = d.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) delta += to_f32(d.e[j]) * to_f32(o.e[j]);
            }
        };
        if (DC == AD && p.D == AD) load_rows(std::false_type{});
        else load_rows(std::true_type{});
    }
    delta += __shfl_xor(delta, 32, 64);
    const int64_t stat_idx = ((int64_t)b * p.Hq + head) * p.lse_st + q_ld;
    const float lse2 = p.LSE[stat_idx] * 1.4426950408889634f;
    if (lh == 0 && q_pos < T_) {
        p.Delta[stat_idx] = -delta;                                       // plane 0: -Delta, the C operand of attn_bwd_dkdv4_kernel's dP MFMAs
        p.Delta[(int64_t)p.B * p.Hq * p.lse_st + stat_idx] = lse2;        // plane 1: LSE log2(e)
    }
    const float delta_s = delta * p.scale;
    // Plain-causal build (which has the registers): -Delta rides as the C operand of the dP chain's first MFMA (16 registers
    // holding the lane's value) and the softmax scale is applied to dQ once, at the store -- dS = P dP' is one multiply per
    // element instead of a fused multiply-add plus a multiply (MFMA time and VALU time add up on this chip).
    constexpr bool CFOLD = !BAND;
    f32x16_t ndelta;
#pragma unroll
    for (int r = 0; r < 16; ++r) ndelta[r] = -delta;
    const int lo_q = BAND ? p.lo[(int64_t)b * T_ + q_ld] : 0;
    const int lo_w0 = BAND ? __builtin_amdgcn_readfirstlane(lo_q) : 0, lo_w1 = BAND ? __builtin_amdgcn_readlane(lo_q, 31) : 0;
    const int t_first = BAND ? p.lo[(int64_t)b * T_ + min(qtile * QT, T_ - 1)] / KT : 0;
    const bool full = BAND && p.noncausal;           // non-causal inside documents: keys lo[q] .. hi[q]
    const int lim_q = full ? p.hi[(int64_t)b * T_ + q_ld] : q_pos;
    const int lim_w0 = full ? __builtin_amdgcn_readfirstlane(lim_q) : qs;
    const int lim_w1 = full ? __builtin_amdgcn_readlane(lim_q, 31) : min(qs + 31, T_ - 1);
    const int lim_blk = full ? p.hi[(int64_t)b * T_ + min(qtile * QT + QT - 1, T_ - 1)] : qtile * QT + QT - 1;

    const int nkv_blk = min(lim_blk / KT + 1, (T_ + KT - 1) / KT);
    const int kvr = p.kvm == 1 ? kvh : kvh / p.kvm;                                     // the K / V head behind a virtual one
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvr * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvr * p.v_sh;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 2048;
    // the four per-lane DMA source offsets are RECOMPUTED at every issue from an opaque copy of the lane id (see
    // attn_fwd_ps_kernel): kept in registers across the tile loop the band build spilled them, and every tile's reload -- a
    // scratch (VMEM) load with a compiler-inserted vmcnt(0) behind it -- drained the LDS-DMA ring (13 spilled dwords, reloaded in
    // all three tile loops, before round 4)
    auto issue = [&](int t, int stage) {
        const int k0 = t * KT;
        const unsigned d = dst_w + stage * STAGE_B;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int rmax = T_ - 1 - k0;                                  // ragged last tile: rows past the end re-read the last key
        unsigned ko[2], vo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + (ln >> 4);
            const int r = min(row, rmax);
            const int sw = slot_in((ln & 15) ^ swz_c(row), p.D) * 16;
            ko[i] = (unsigned)(r * (int)p.k_st * 2 + sw);
            vo[i] = (unsigned)(r * (int)p.v_st * 2 + sw);
        }
        dma16x2(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], d, d + 1024);
        dma16x2(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], d + TILE_B, d + TILE_B + 1024);
    };

    // Plain-causal build (which has the registers): whole tiles take a short issue -- constant per-lane offsets, running K / V
    // pointers, a counted ring stage (see attn_fwd_ps_kernel: ~85 -> ~30 instructions per tile step)
    unsigned kof[2] = {0, 0}, vof[2] = {0, 0};
    if constexpr (!BAND) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + (lane >> 4);
            const int sw = slot_in((lane & 15) ^ swz_c(row), p.D) * 16;
            kof[i] = (unsigned)(row * (int)p.k_st * 2 + sw);
            vof[i] = (unsigned)(row * (int)p.v_st * 2 + sw);
        }
    }
    const int64_t kstep = (int64_t)KT * p.k_st, vstep = (int64_t)KT * p.v_st;
    const T* kq = kbase + (int64_t)(t_first + 2) * kstep;          // K / V rows of tile ti + 2 (running)
    const T* vq = vbase + (int64_t)(t_first + 2) * vstep;
    int stg = 0;                                                    // ring stage of the current tile = ti % 3, counted

    // row reads (K for S^T, V for dP^T): lane -> row l31 (+32 kt), 16 B at logical slot 2 ks + lh
    const int r_lane = l31 * 256 + ((swz_c(l31 & 15) ^ lh) << 4);
    // transposing reads of K (A operand of dQ^T), swizzle C: second 4-row block = (addr ^ 32) + 8 rows
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int t_lane = (4 * lh + (sg >> 2)) * 256 +
                       ((((sg >> 2) << 2) | (((gh << 1) | ((sg >> 1) & 1)) ^ lh)) << 4) + (sg & 1) * 8;

    f32x16_t dq_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_acc[i][r] = 0.f;

    const int nt = nkv_blk - t_first;
    issue(t_first, 0);
    if (nt > 1) issue(t_first + 1, 1);
    const int last_tile_wave = lim_w1 / KT;
    const int first_tile_wave = lo_w0 / KT;
    // One tile step. MASKED is a compile-time flag and the tile range is split by hand into
    // [band-edge tiles | interior tiles | diagonal / ragged tiles]: with a run-time `need_mask` that depends on
    // the band the compiler keeps both paths' registers alive in one loop body (+50 VGPRs, spills).
    // (DC: the head-dim class, see attn_fwd_kernel -- k-steps of S^T / dP^T and d-tiles of dQ^T past it are not computed)
    auto step = [&](int ti, auto mode_c, bool rt_mask) {       // mode 0: no mask, 1: mask, 2: mask iff rt_mask
        constexpr int MODE = decltype(mode_c)::value;
        constexpr int NKS = DC / 16, NDT = DC / 32;
        const int t = t_first + ti;
        if (ti + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int stg_cur = stg, stg2 = stg == 0 ? 2 : stg - 1;        // ti % 3, (ti + 2) % 3
        stg = stg == NST - 1 ? 0 : stg + 1;
        if (ti + 2 < nt) {
            if (!BAND && (t + 3) * KT <= T_) {
                const unsigned d = dst_w + stg2 * STAGE_B;
                dma16x2(kq, kof[0], kof[1], d, d + 1024);
                dma16x2(vq, vof[0], vof[1], d + TILE_B, d + TILE_B + 1024);
            } else {
                issue(t + 2, stg2);
            }
            kq += kstep;
            vq += vstep;
        }
        if (t > last_tile_wave || t < first_tile_wave) return;
        const unsigned char* sk = smem + stg_cur * STAGE_B;
        const unsigned char* sv = sk + TILE_B;
        const int k0 = t * KT;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16_t st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
            if constexpr (CFOLD) dp = ndelta;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                union { uint4 r; frag_t f; } u, w;
                u.r = *reinterpret_cast<const uint4*>(sk + kt * 32 * 256 + (r_lane ^ (ks * 32)));
                w.r = *reinterpret_cast<const uint4*>(sv + kt * 32 * 256 + (r_lane ^ (ks * 32)));
                st = MfmaA<T>::run(u.f, qf[ks], st);
                dp = MfmaA<T>::run(w.f, dof[ks], dp);
            }
            // dS^T = P^T (dP^T - Delta) * scale, masked entries 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], p.scale_log2, -lse2));
                if (MODE == 1 || (MODE == 2 && rt_mask)) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > lim_q || key >= T_ || key < lo_q) pv = 0.f;
                }
                if constexpr (CFOLD) st[r] = pv * dp[r];
                else st[r] = pv * __builtin_fmaf(dp[r], p.scale, -delta_s);   // (dP - Delta) * scale in one fma
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                union { uint32_t w[4]; frag_t f; } sb;
#pragma unroll
                for (int j = 0; j < 4; ++j) sb.w[j] = BAND ? pack_pair<T>(st[8 * c + 2 * j], st[8 * c + 2 * j + 1])   // (the packed form spills in the band build)
                                       : pack_pair2<T>(st[8 * c + 2 * j], st[8 * c + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const int a0 = (kt * 32 + c * 16) * 256 + (t_lane ^ (dt << 6));
                    union { s16x4_t h[2]; frag_t f; } ka;
                    ka.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sk + a0));
                    ka.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sk + (a0 ^ 32) + 8 * 256));
                    dq_acc[dt] = MfmaA<T>::run(ka.f, sb.f, dq_acc[dt]);
                }
            }
        }
    };
    {
        // tiles whose first key lies below the wave's largest band edge: a prefix; tiles that touch the diagonal
        // or run past T: a suffix
        const int t_pre_end = min(nkv_blk, (lo_w1 + KT - 1) / KT);
        const int t_diag = (lim_w0 + 1) / KT, t_rag = (T_ % KT) ? T_ / KT : nkv_blk;
        const int t_suf = max(t_pre_end, min(min(t_diag, t_rag), nkv_blk));
        if constexpr (BAND) {
            int ti = 0;
            for (; t_first + ti < t_pre_end; ++ti) step(ti, std::integral_constant<int, 1>{}, true);
            for (; t_first + ti < t_suf; ++ti) step(ti, std::integral_constant<int, 0>{}, false);
            for (; ti < nt; ++ti) step(ti, std::integral_constant<int, 1>{}, true);
        } else {
            // plain causal: ONE loop with the (monotone) mask test around the masking statement only -- hipcc
            // splits the range itself and needs far fewer registers than with the hand-split loops
            for (int ti = 0; ti < nt; ++ti) step(ti, std::integral_constant<int, 2>{}, ti >= t_suf);
        }
    }
    {
        int qr = q_ld;                       // (the row address formed HERE: hoisted, the pointer pair is spilled around the loop)
        asm volatile("" : "+v"(qr));
        T* op = (T*)p.dQ + b * p.dq_sb + (int64_t)qr * p.dq_st + (int64_t)head * p.dq_sh;
        if (p.D == AD) store_rows_x4<T>(op, dq_acc, CFOLD ? p.scale : 1.0f, lh, q_pos < T_);
        else store_rows_x4<T>(op, dq_acc, CFOLD ? p.scale : 1.0f, lh, q_pos < T_, p.D);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Backward, part 2, round-3 structure: dK, dV with ONE wave per SIMD.
// What was wrong with attn_bwd_dkdv_kernel (8 waves x 256 registers, profiles/r02zz_pmc_mfma_busy.md: MFMA-busy 21 %,
// 9,500 cycles per 2,048-cycle step): 128 accumulator registers + 32 of K^T + 32 of scores leave no room to read
// operands ahead, so hipcc emits `ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma` 32 times per step (every MFMA pays a full
// LDS round trip), K^T is re-loaded from L2 at the top of every step behind a vmcnt the whole block waits on, every
// step ends in a full-block `vmcnt(0) + s_barrier`, and each wave reads Q, dO, Q^T, dO^T from LDS for only 32 keys.
// Here a block is 4 waves = the 4 units of the old kernel (query heads of the group / q slices), each wave owns BOTH
// 32-key halves of the block's 64 keys and the whole 512-entry register file of its SIMD:
//   * accumulators dK^T, dV^T for 64 keys = 256 AGPRs a0..a255, touched only by inline asm naming the physical
//     registers (attn_acc256.inc; with compiler-managed accumulators hipcc copies tuples between the files at every
//     control-flow merge and spills), K^T fragments resident in 64 VGPRs for the whole kernel (no reload), V in LDS
//     (row reads, shared by the 4 waves);
//   * every Q / dO / Q^T / dO^T fragment read from LDS feeds TWO MFMAs (one per key half): 0.75 KB of LDS reads per
//     MFMA instead of 1.25;
//   * the Q / dO staging ring is PRIVATE to the wave (2 stages x 16 KiB + a 256-byte line of LSE2 / Delta, filled by
//     the wave's own LDS-DMA): a step starts on the wave's own `vmcnt(0)` -- no block barrier anywhere in the loop,
//     the four SIMDs drift freely;
//   * a step is a hand-pipelined stream of 32 chunks of two MFMAs (see `body`), each MFMA group with independent VALU
//     work beside it:
//       S = Q K^T (16 MFMA)            | operand prefetch
//       dP = dO V^T (16 MFMA)          | P = exp2(S c - LSE2), packed to 16 bit; the next step's LDS-DMA
//       dV^T += dO^T P (16 MFMA)       | dS' = P (dP - Delta), packed  (the softmax scale is applied to dK^T once, at the end)
//       dK^T += Q^T dS' (16 MFMA)      |
// LSE2 = LSE log2(e) is written next to Delta by attn_bwd_dq_kernel (plane 1 of the Delta scratch). Same Q / dO / V
// LDS tile formats and swizzles as the old kernel; same fixed-order reduction of the 4 units at the end.
// LDS map: [V tile 16 KiB][stats 2 KiB: (stage, unit) x (32 LSE2 | 32 Delta)][ring: unit x stage x (Q 8 KiB | dO 8 KiB)]
// -- the ring is unit-major so that stage / operand / k-step select an IMMEDIATE offset (< 64 KiB) on a per-lane constant.
#ifndef UAMD_KD4_DMA_CHUNK
#define UAMD_KD4_DMA_CHUNK 0      // first of the five chunks that carry the next step's LDS-DMA (0: beside the S MFMAs)
#endif
#ifndef UAMD_KD4_PF
#define UAMD_KD4_PF 2             // operand prefetch distance in chunks (LDS round trip vs 64 MFMA cycles per chunk)
#endif
constexpr int KD4_STATS_OFF = TILE_B;                 // 16 KiB
constexpr int KD4_RING_OFF = KD4_STATS_OFF + 2048;    // 18 KiB
constexpr int KD4_LDS = KD4_RING_OFF + 4 * 32768;     // 149,504 B

// one LDS-DMA wave-instruction of 4 bytes per lane: lane l copies [gptr_l, +4) to LDS [dst + 4 l)
__device__ __forceinline__ void dma4x1(const void* gptr, unsigned d0) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(d0)
        : "memory");
}

// compile-time loop: f(std::integral_constant<int, K>{}) for K = 0 .. N - 1 (every index a constant expression: register
// arrays indexed with it never go to scratch, `if constexpr` chains fold)
template <typename F, int... Ks>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Ks...>) {
    (f(std::integral_constant<int, Ks>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// score MFMAs with the accumulator in VGPRs (the softmax reads it), as inline asm: with the builtin hipcc parks these
// accumulators in AGPRs a0..a63 between the asm statements -- on top of the pinned dV^T tuples. The wait states a VALU
// reader needs behind the chain's last MFMA are provided by the schedule (>= 2 MFMA issue slots, see the kernel).
template <typename T>
__device__ __forceinline__ void vmfma_first(f32x16_t& s, typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
    if constexpr (std::is_same<T, bf16_t>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(a), "v"(b));
}
template <typename T>
__device__ __forceinline__ void vmfma(f32x16_t& s, typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
    if constexpr (std::is_same<T, bf16_t>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(a), "v"(b));
}

#include "attn_acc256.inc"
#include "attn_kd4_loop.inc"
// tuple I of the accumulator file -> this wave's slab of the reduction buffer: red[unit][(I & 7) * 16 + r][lane]
template <int I>
__device__ __forceinline__ void acc256_to_lds(float* red, int unit, int lane) {
    float f[16];
    acc256_read<I>(f);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(unit * 128 + (I & 7) * 16 + r) * 64 + lane] = f[r];
}

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) attn_bwd_dkdv4_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int unit = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    // The G query heads of the KV head go through the four waves in PASSES of 4, 2 or 1 heads -- G = 4 a + 2 b + c: a passes of
    // a head per wave, then (b) two heads x two q-slices, then (c) one head x four q-slices (q-slice: every nslice-th 32-row
    // step of the head's queries) -- so no wave idles whatever the group size (3, 5, 6, 7: Llama-3.2-3B, Qwen2.5-14B / 32B,
    // Qwen2.5-7B / Qwen2-VL-7B = BASELINE config 4; rounds 2-5 ran them zero-padded to 4 / 8 heads through copies of Q, O, dO).
    const int npass = (G >> 2) + ((G >> 1) & 1) + (G & 1);
    int hpp, nslice, hin, slice, head0;               // of the current pass (set at its top)
    const int npairs = p.Hk * p.B;
    int jt, pair_;                                    // key tile 0 sees every q tile (causal): heaviest first
    block_to_work((int)blockIdx.x, npairs, jt, pair_);
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int k0 = jt * KT;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    // band upper edge: last query that attends a key (non-decreasing in key); per key half
    int hi_k[2], key[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        key[kh] = k0 + kh * 32 + l31;
        const int key_ld = key[kh] < T_ ? key[kh] : T_ - 1;
        hi_k[kh] = p.hi ? p.hi[(int64_t)b * T_ + key_ld] : T_ - 1;
    }
    const int hi_w0 = __builtin_amdgcn_readfirstlane(hi_k[0]);                   // smallest edge of the tile
    const int hi_blk = p.hi ? p.hi[(int64_t)b * T_ + min(k0 + KT - 1, T_ - 1)] : T_ - 1;
    // lower edge: the key itself (causal: queries before the key do not see it) or -- non-causal attention inside documents --
    // the first query of the key's document
    const bool full = p.noncausal != 0;
    int qlo_k[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int key_ld = key[kh] < T_ ? key[kh] : T_ - 1;
        qlo_k[kh] = full ? p.lo[(int64_t)b * T_ + key_ld] : key[kh];
    }
    const int qlo_blk = full ? p.lo[(int64_t)b * T_ + k0] : k0;                                  // smallest lower edge of the tile
    const int qlo_w1 = full ? p.lo[(int64_t)b * T_ + min(k0 + KT - 1, T_ - 1)] : k0 + KT - 1;    // largest

    // ---- V tile -> LDS (swizzle C), 16 pieces of 1 KiB, 4 per wave
    {
        const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh + (int64_t)k0 * p.v_st;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (unit * 4 + i) * 4 + (lane >> 4);
            const int r = min(row, T_ - 1 - k0);
            dma16x1(vbase + (int64_t)r * p.v_st + slot_in((lane & 15) ^ swz_c(row), p.D) * 8, lds_base + (unit * 4 + i) * 1024);
        }
    }
    // ---- K^T operand (B of S = Q K^T: lane -> key, 8 d at 16 ks + 8 lh), both key halves, resident
    frag_t kf[2][8];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int key_ld = key[kh] < T_ ? key[kh] : T_ - 1;
        const T* kp = (const T*)p.K + b * p.k_sb + (int64_t)key_ld * p.k_st + (int64_t)kvh * p.k_sh + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bool in = ks * 16 + lh * 8 < p.D;                    // (head dims below 128: zero past D; unconditional loads)
            union { uint4 r; frag_t f; } u;
            const unsigned keep = in ? 0xffffffffu : 0u;
            u.r = *reinterpret_cast<const uint4*>(in ? kp + ks * 16 : kp - lh * 8);
            u.r.x &= keep; u.r.y &= keep; u.r.z &= keep; u.r.w &= keep;
            kf[kh][ks] = u.f;
        }
    }

    // DMA source offsets of the wave's Q and dO tiles: piece i (rows 4 i + (lane >> 4)) = base(+16 rows for i >= 4)
    // + row * stride * 2 + (dsw0 ^ ((i & 3) << 4))   (swz_c(row) = ((lane >> 4) << 2) | (i & 3) for these rows)
    // (head dims below 128: a lane whose slot lies past D re-reads slot 0 of its row -- the Q / dO columns past D meet zero K / V)
    const int dsl0 = (lane & 15) ^ ((lane >> 4) << 2);
    unsigned dsw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dsw[i] = (unsigned)(slot_in(dsl0 ^ i, p.D) << 4);
    unsigned qo[4], doo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qo[i] = (unsigned)((int64_t)(i * 4 + (lane >> 4)) * p.q_st * 2) + dsw[i];
        doo[i] = (unsigned)((int64_t)(i * 4 + (lane >> 4)) * p.do_st * 2) + dsw[i];
    }
    const int nq32 = (T_ + 31) / 32;
    const int q32_first = qlo_blk / 32;
    int nsteps;                                        // steps of the current pass: the band's 32-row q tiles / nslice

    // ---- per-lane ABSOLUTE LDS byte addresses (swizzle C; the dynamic region's base included), made opaque once:
    //      stage / operand / k-step / c are immediates on them
    const unsigned ring_u = lds_base + KD4_RING_OFF + unit * 32768;
    const int r_lane = l31 * 256 + ((swz_c(l31 & 15) ^ lh) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int t_lane = (4 * lh + (sg >> 2)) * 256 +
                       ((((sg >> 2) << 2) | (((gh << 1) | ((sg >> 1) & 1)) ^ lh)) << 4) + (sg & 1) * 8;
    unsigned cq[8], cv[8], ct[4], ct2[4];              // row reads of the ring / of V; transposing reads (two 4-row blocks)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        cv[ks] = lds_base + (unsigned)(r_lane ^ (ks * 32));
        cq[ks] = ring_u + (unsigned)(r_lane ^ (ks * 32));
        asm volatile("" : "+v"(cq[ks]), "+v"(cv[ks]));
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        ct[dt] = ring_u + (unsigned)(t_lane ^ (dt << 6));
        ct2[dt] = ring_u + (unsigned)((t_lane ^ (dt << 6)) ^ 32) + 8 * 256;
        asm volatile("" : "+v"(ct[dt]), "+v"(ct2[dt]));
    }
    unsigned cs = lds_base + KD4_STATS_OFF + unit * 256 + lh * 16;   // stats line: + stage * 1024 + quad * 32 + (pair in quad) * 8 (+ 128: -Delta)
    asm volatile("" : "+v"(cs));
    // the stats line's DMA source of the generated loop: lanes 0-31 LSE2 (plane 1), lanes 32-63 -Delta (plane 0), relative to the
    // -Delta row (host: one plane < 2^31 bytes)
    const unsigned vstat = (lh ? 0u : (unsigned)(p.B * p.Hq * p.lse_st) * 4u) + (unsigned)l31 * 4u;
    const bool no_asm = p.no_asm != 0;                 // (A/B and parity tests: every step through the C++ body)
    // the masked loop's band edges per key half, relative to the lane half's first row: key k attends rows qlo .. qhi
    int mqlo[2], mqhi[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        mqlo[kh] = qlo_k[kh] - 4 * lh;
        mqhi[kh] = min(hi_k[kh], T_ - 1) - 4 * lh;
    }

    // (the accumulators are zeroed and the V tile awaited inside the first pass, BEHIND the issue of its first Q / dO tile: one
    // DMA round trip per workgroup less on the critical path)
    const float* lse2_all = p.Delta + (int64_t)p.B * p.Hq * p.lse_st;            // plane 1 of the scratch: LSE * log2(e)
    for (int pass = 0; pass < npass; ++pass) {
        if (pass < (G >> 2)) { hpp = 4; head0 = 4 * pass; }
        else if ((G & 2) && pass == (G >> 2)) { hpp = 2; head0 = G & ~3; }
        else { hpp = 1; head0 = G - 1; }
        const int hsh = hpp == 4 ? 2 : hpp - 1;          // log2(hpp)
        nslice = 4 >> hsh;
        hin = unit & (hpp - 1);
        slice = unit >> hsh;
        nsteps = (min(nq32, hi_blk / 32 + 1) - q32_first + nslice - 1) >> (2 - hsh);
        const int head = kvh * G + head0 + hin;
        const T* qbase = (const T*)p.Q + b * p.q_sb + (int64_t)head * p.q_sh;
        const T* dobase = (const T*)p.dO + b * p.do_sb + (int64_t)head * p.do_sh;
        const float* lse_row = lse2_all + ((int64_t)b * p.Hq + head) * p.lse_st;
        const float* del_row = p.Delta + ((int64_t)b * p.Hq + head) * p.lse_st;
        // the steps of a pass sweep the query tiles DOWNWARDS, from the band's upper end to the diagonal: whatever their key
        // tile, the workgroups that run side by side on an XCD (the heaviest-first order starts them together) then stream the
        // SAME Q / dO rows at the same time -- one L2 fill serves them all. Upwards from each block's own diagonal, a row was
        // re-read two steps after its first use, with the XCD's whole traffic of two steps (4 MB = its L2) in between.
#ifdef UAMD_KD4_UPWARD
        auto q0_of = [&](int step) { return (q32_first + step * nslice + slice) * 32; };
        constexpr int SWEEP = 1;
#else
        auto q0_of = [&](int step) { return (q32_first + (nsteps - 1 - step) * nslice + slice) * 32; };
        constexpr int SWEEP = -1;
#endif
        // tile a step's DMA fetches: its own q tile, or (idle slice at the end of the sequence) the last valid one
        auto fetch_q0 = [&](int step) {
            const int q0 = q0_of(step);
            return q0 >= T_ ? (nq32 - 1) * 32 : q0;
        };
        // part: -1 = everything, 0 / 1 = Q rows 0-15 / 16-31, 2 / 3 = dO rows 0-15 / 16-31, 4 = the LSE2 | Delta line.
        // FULL: the tile has all 32 rows (precomputed offsets, no branch); else rows past the end re-read the last row.
        auto issue = [&](int q0, int stage, int part, auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            const unsigned d = lds_base + KD4_RING_OFF + unit * 32768 + stage * 16384;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (!(part < 0 || part == h)) continue;
                const T* base = h < 2 ? qbase : dobase;
                const int64_t st_ = h < 2 ? p.q_st : p.do_st;
                if (FULL) {
                    const unsigned* o = h < 2 ? qo : doo;
                    dma16x4g(base + (int64_t)(q0 + (h & 1) * 16) * st_, o[0], o[1], o[2], o[3], d + h * 4096);
                } else {
                    unsigned o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = min(((h & 1) * 4 + i) * 4 + (lane >> 4), T_ - 1 - q0);
                        o[i] = (unsigned)((int64_t)r * st_ * 2) + dsw[i];
                    }
                    dma16x4g(base + (int64_t)q0 * st_, o[0], o[1], o[2], o[3], d + h * 4096);
                }
            }
            // LSE2 (lanes 0-31) | Delta (lanes 32-63) of rows q0 .. q0 + 31 (the stat rows are padded to a multiple of 32)
            if (part < 0 || part == 4)
                dma4x1((lh ? del_row : lse_row) + q0 + l31, lds_base + KD4_STATS_OFF + (stage * 4 + unit) * 256);
        };

        // The step is a stream of 32 CHUNKS of two MFMAs (one LDS operand fragment x the two key halves):
        //   0-7   S[kh]   += Q[:, ks] K^T[kh]          operand: Q rows (1 b128 read)
        //   8-15  dP[kh]  += dO[:, ks] V^T[kh]         operands: dO rows, V rows of both halves (3 reads)
        //   16-23 dV^T[kh][dt] += dO^T[dt, c] P[kh][c] operand: dO^T (2 transposing reads)
        //   24-31 dK^T[kh][dt] += Q^T[dt, c] dS[kh][c] operand: Q^T
        // software-pipelined by hand: chunk k + 2's operands are requested before chunk k's MFMAs (every MFMA here is
        // inline asm -- the accumulators are registers hipcc must not touch -- so the compiler cannot see what a fragment
        // feeds), and the VALU work rides beside MFMAs that do not depend on it:
        //   chunks 9-15  P = exp2(S c - LSE2) and its packing, pair by pair (S is complete after chunk 7)
        //   chunks 17-23 dS' = P (dP - Delta) and its packing (dP is complete after chunk 15)
        //   chunks 0-4   the next step's LDS-DMA (4 x 4 KiB of Q / dO + the stats line), beside the S MFMAs that have no VALU
        //                work of their own (UAMD_KD4_DMA_CHUNK)
        // A VALU read of an MFMA result is always >= 2 MFMA issue slots behind the chain's last MFMA (the XDL write -> VALU
        // read wait states are covered by the MFMAs in between). Placement is pinned with EMPTY volatile asm statements
        // (volatile asms keep their order, and every MFMA is one): inputs are made opaque AFTER the chunk's MFMAs -- a pure
        // fp expression would otherwise float up to right behind the MFMA chain that produces its operand -- and results
        // are consumed by an empty asm BEFORE the next chunk's MFMAs, so a packed operand is never written right in front
        // of the (asm) MFMA that reads it.
        auto body = [&](auto stage_c, int q0, int qn) {
            constexpr bool MASK = true;        // (the unmasked steps of a pass run in the generated loop, KD4_LOOP)
            constexpr int STAGE = decltype(stage_c)::value;
            constexpr int SO = STAGE * 16384;
            f32x16_t sc[2], dp[2];
            union { uint32_t w[8]; frag_t f[2]; } pb[2], sb[2];   // P / dS' as B operands: f[c] = rows 16 c .. 16 c + 15
            constexpr int PF = UAMD_KD4_PF;
            frag_t ob[PF + 1][3];
            float2 st2[2][3];                                     // stats of the pairs of the next chunk (ping-pong)
            auto rd128 = [&](unsigned addr) {
                union { u32x4a_t r; frag_t f; } u;
                u.r = *(const lds_u32x4a*)(uintptr_t)addr;
                return u.f;
            };
            auto rdtr = [&](unsigned a0, unsigned a1) {
                union { s16x4_t h[2]; frag_t f; } t;
                t.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a0);
                t.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a1);
                return t.f;
            };
            auto reads = [&](auto kc, frag_t* o) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < 8) {
                    o[0] = rd128(cq[k] + SO);
                } else if constexpr (k < 16) {
                    constexpr int ks = k - 8;
                    o[0] = rd128(cq[ks] + (SO + 8192));
                    o[1] = rd128(cv[ks]);
                    o[2] = rd128(cv[ks] + 32 * 256);
                } else if constexpr (k < 24) {
                    constexpr int c = (k - 16) >> 2, dt = (k - 16) & 3;
                    o[0] = rdtr(ct[dt] + (SO + 8192 + c * 4096), ct2[dt] + (SO + 8192 + c * 4096));
                } else {
                    constexpr int c = (k - 24) >> 2, dt = (k - 24) & 3;
                    o[0] = rdtr(ct[dt] + (SO + c * 4096), ct2[dt] + (SO + c * 4096));
                }
            };
            auto mfmas = [&](auto kc, const frag_t* o, auto half_c) {     // half = key half: the chunk's first / second MFMA
                constexpr int k = decltype(kc)::value, kh = decltype(half_c)::value;
                if constexpr (k < 8) {
                    if constexpr (k == 0) vmfma_first<T>(sc[kh], o[0], kf[kh][0]);
                    else vmfma<T>(sc[kh], o[0], kf[kh][k]);
                } else if constexpr (k < 16) {
                    vmfma<T>(dp[kh], o[0], o[1 + kh]);             // (dp starts at -Delta, below)
                } else if constexpr (k < 24) {
                    constexpr int c = (k - 16) >> 2, dt = (k - 16) & 3;
                    acc256_mfma<T, 4 * kh + dt>(o[0], pb[kh].f[c]);
                } else {
                    constexpr int c = (k - 24) >> 2, dt = (k - 24) & 3;
                    acc256_mfma<T, 8 + 4 * kh + dt>(o[0], sb[kh].f[c]);
                }
            };
            // pair pi = 2 j + kh (j-major: the c = 0 operands of both halves are complete first): registers 2 j, 2 j + 1 =
            // q rows q0 + 8 (j >> 1) + 4 lh + 2 (j & 1) + {0, 1}; their LSE2 (Delta: + 128 bytes) sit side by side
            auto stat_pair = [&](auto pic, int delta) {
                constexpr int j = decltype(pic)::value >> 1;
                const f32x2a_t v = *(const lds_f32x2a*)(uintptr_t)(cs + (STAGE * 1024 + (j >> 1) * 32 + (j & 1) * 8) + delta);
                return make_float2(v[0], v[1]);
            };
            auto p_pair = [&](auto pic, float2 l2) {
                constexpr int pi = decltype(pic)::value, kh = pi & 1, j = pi >> 1;
                const float lv[2] = {l2.x, l2.y};
                float x[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kh][2 * j + e], p.scale_log2, -lv[e]));
                    if (MASK) {
                        const int r = 2 * j + e;
                        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (q < qlo_k[kh] || q >= T_ || key[kh] >= T_ || q > hi_k[kh]) pv = 0.f;
                    }
                    x[e] = pv;
                    sc[kh][2 * j + e] = pv;
                }
                pb[kh].w[j] = pack_pair2<T>(x[0], x[1]);
                return pb[kh].w[j];
            };
            auto ds_pair = [&](auto pic) {                       // masked entries have P = 0, hence dS' = 0
                constexpr int pi = decltype(pic)::value, kh = pi & 1, j = pi >> 1;
                const float x0 = sc[kh][2 * j] * dp[kh][2 * j];
                const float x1 = sc[kh][2 * j + 1] * dp[kh][2 * j + 1];
                sb[kh].w[j] = pack_pair2<T>(x0, x1);
                return sb[kh].w[j];
            };
            // VALU work of chunk k, in two halves: `first` runs between the chunk's two MFMAs, the rest behind the second.
            // 16 pairs over 7 chunks: 3 3 2 2 2 2 2 (slot -> first pair, count); one pair in the first half.
            auto valu = [&](auto kc, auto half_c) {
                constexpr int k = decltype(kc)::value;
                constexpr bool FIRST = decltype(half_c)::value == 0;
                if constexpr (k >= 9 && k < 16) {
                    constexpr int slot = k - 9, first = slot < 2 ? 3 * slot : 6 + 2 * (slot - 2), count = slot < 2 ? 3 : 2;
                    if constexpr (FIRST) {
                        // S is complete (its last MFMA is >= 3 MFMA slots back): pin the reads of it BEHIND this point
                        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
                        const uint32_t w0 = p_pair(std::integral_constant<int, first>{}, st2[k & 1][0]);
                        asm volatile("" :: "v"(w0));
                    } else {
                        const uint32_t w1 = p_pair(std::integral_constant<int, first + 1>{}, st2[k & 1][1]);
                        uint32_t w2 = w1;
                        if constexpr (count == 3) w2 = p_pair(std::integral_constant<int, first + 2>{}, st2[k & 1][2]);
                        asm volatile("" :: "v"(w1), "v"(w2));
                    }
                } else if constexpr (k >= 17 && k < 24) {
                    constexpr int slot = k - 17, first = slot < 2 ? 3 * slot : 6 + 2 * (slot - 2), count = slot < 2 ? 3 : 2;
                    if constexpr (FIRST) {
                        asm volatile("" : "+v"(dp[0]), "+v"(dp[1]));
                        const uint32_t w0 = ds_pair(std::integral_constant<int, first>{});
                        asm volatile("" :: "v"(w0));
                    } else {
                        const uint32_t w1 = ds_pair(std::integral_constant<int, first + 1>{});
                        uint32_t w2 = w1;
                        if constexpr (count == 3) w2 = ds_pair(std::integral_constant<int, first + 2>{});
                        asm volatile("" :: "v"(w1), "v"(w2));
                    }
                }
                if constexpr (!FIRST) {
                    // LSE2 of the pairs chunk k + 1 will process (P: chunks 9-15)
                    if constexpr (k >= 8 && k < 15) {
                        constexpr int slot = k - 8, first = slot < 2 ? 3 * slot : 6 + 2 * (slot - 2);
                        constexpr int count = slot < 2 ? 3 : 2;
                        st2[(k + 1) & 1][0] = stat_pair(std::integral_constant<int, first>{}, 0);
                        st2[(k + 1) & 1][1] = stat_pair(std::integral_constant<int, first + 1>{}, 0);
                        if constexpr (count == 3) st2[(k + 1) & 1][2] = stat_pair(std::integral_constant<int, first + 2>{}, 0);
                    }
                    if constexpr (k >= UAMD_KD4_DMA_CHUNK && k < UAMD_KD4_DMA_CHUNK + 5)
                        issue(qn, STAGE ^ 1, k - UAMD_KD4_DMA_CHUNK, std::false_type{});
                }
            };
            // dP starts at -Delta[q] (plane 0 of the scratch holds the negated row sums): dP' = dO V^T - Delta comes out of the
            // MFMA chain, the same arithmetic as the generated loop's
#pragma unroll
            for (int g_ = 0; g_ < 4; ++g_) {
                const u32x4a_t v = *(const lds_u32x4a*)(uintptr_t)(cs + (STAGE * 1024 + 128 + g_ * 32));
#pragma unroll
                for (int e = 0; e < 4; ++e) dp[0][4 * g_ + e] = dp[1][4 * g_ + e] = __uint_as_float(v[e]);
            }
            static_for<PF>([&](auto kc) { reads(kc, ob[decltype(kc)::value]); });
            static_for<32>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k + PF < 32) reads(std::integral_constant<int, k + PF>{}, ob[(k + PF) % (PF + 1)]);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kc, ob[k % (PF + 1)], std::integral_constant<int, 0>{});
                valu(kc, std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kc, ob[k % (PF + 1)], std::integral_constant<int, 1>{});
                valu(kc, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto run = [&](auto stage_c, int step) {
            constexpr int STAGE = decltype(stage_c)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's tiles of `step` are in LDS
            // the DMA issued DURING this step: the next step's tile, or -- last step -- this step's tile once more into the
            // idle stage (unconditional: no branch inside the chunk stream; drained before the epilogue reuses the LDS)
            const int qn = fetch_q0(step + 1 < nsteps ? step + 1 : step);
            const int q0 = q0_of(step);
            if (q0 >= T_ || q0 > hi_blk) {                        // idle slice / below the band
                issue(qn, STAGE ^ 1, -1, std::false_type{});
                return;
            }
            body(stage_c, q0, qn);
        };
        // Steps over WHOLE tiles (32 query rows and 64 keys inside the sequence, and a whole tile to prefetch) run in the generated
        // loops (attn_kd4_loop.inc, tools/gen/gen_attn_kd4.py): consecutive steps of one class are ONE asm statement that alternates
        // the ring stages itself, prefetches step s + 1's tile during step s and advances its sources by a constant. Class 1: no
        // element masked. Class 2: diagonal / band-edge tiles -- the valid rows of a key are an interval, turned into a bit mask
        // per step and ANDed into P (dS' = P dP' follows). Class 0 (ragged tiles, idle slices): the C++ body above. The last step
        // of a pass prefetches its own tile once more (a spare, like the C++ path), so it is a run of its own.
        auto cls = [&](int s_) {
            const int q0 = q0_of(s_);
            if (q0 + 32 > T_ || k0 + KT > T_ || q0 > hi_blk) return 0;
            const int qn = s_ + 1 < nsteps ? q0_of(s_ + 1) : q0;
            if (qn + 32 > T_) return 0;
            return (q0 < qlo_w1 || q0 + 31 > hi_w0) ? 2 : 1;
        };
        auto asm_run = [&](bool masked, int step, int n) {
            const int qn = step + 1 < nsteps ? q0_of(step + 1) : q0_of(step);
            auto lo32 = [](uint64_t x) { return __builtin_amdgcn_readfirstlane((unsigned)x); };
            auto hi32 = [](uint64_t x) { return __builtin_amdgcn_readfirstlane((unsigned)(x >> 32)); };
            const uint64_t q0p = (uint64_t)(uintptr_t)(qbase + (int64_t)qn * p.q_st), q1p = q0p + (uint64_t)(32 * p.q_st);
            const uint64_t d0p = (uint64_t)(uintptr_t)(dobase + (int64_t)qn * p.do_st), d1p = d0p + (uint64_t)(32 * p.do_st);
            const uint64_t stp = (uint64_t)(uintptr_t)(del_row + qn);
            const uint64_t advq = (uint64_t)((int64_t)SWEEP * nslice * 64 * p.q_st);          // 32 rows x 2 bytes, signed
            const uint64_t advd = (uint64_t)((int64_t)SWEEP * nslice * 64 * p.do_st);
            const uint64_t advs = (uint64_t)((int64_t)SWEEP * nslice * 32 * 4);
            const unsigned ringu = ring_u, statu = lds_base + KD4_STATS_OFF + unit * 256;
            unsigned cnt = (unsigned)(n - 1);
            const unsigned stage = (unsigned)(step & 1);
#define KD4_SGPR_IN                                                                                                              \
            [q0lo] "s"(lo32(q0p)), [q0hi] "s"(hi32(q0p)), [q1lo] "s"(lo32(q1p)), [q1hi] "s"(hi32(q1p)), [d0lo] "s"(lo32(d0p)),  \
            [d0hi] "s"(hi32(d0p)), [d1lo] "s"(lo32(d1p)), [d1hi] "s"(hi32(d1p)), [stlo] "s"(lo32(stp)), [sthi] "s"(hi32(stp)),   \
            [advqlo] "s"(lo32(advq)), [advqhi] "s"(hi32(advq)), [advdlo] "s"(lo32(advd)), [advdhi] "s"(hi32(advd)),             \
            [advs] "s"(lo32(advs)), [advshi] "s"(hi32(advs)), [ringu] "s"(ringu), [statu] "s"(statu), [sl2] "s"(p.scale_log2), [stage] "s"(stage)
            if (!masked) {
                if constexpr (std::is_same<T, bf16_t>::value)
                    asm volatile(KD4_LOOP("bf16") : [cnt] "+s"(cnt) : KD4_IN_KF, KD4_IN_ADDR, KD4_IN_DMA, KD4_SGPR_IN : KD4_CLOBBER);
                else
                    asm volatile(KD4_LOOP("f16") : [cnt] "+s"(cnt) : KD4_IN_KF, KD4_IN_ADDR, KD4_IN_DMA, KD4_SGPR_IN : KD4_CLOBBER);
            } else {
                const unsigned q0s = (unsigned)q0_of(step), rowadv = (unsigned)(SWEEP * nslice * 32);
                if constexpr (std::is_same<T, bf16_t>::value)
                    asm volatile(KD4_LOOP_M("bf16") : [cnt] "+s"(cnt)
                                 : KD4_IN_KF, KD4_IN_ADDR, KD4_IN_DMA, KD4_IN_MASK, KD4_SGPR_IN, [q0s] "s"(q0s), [rowadv] "s"(rowadv) : KD4_CLOBBER);
                else
                    asm volatile(KD4_LOOP_M("f16") : [cnt] "+s"(cnt)
                                 : KD4_IN_KF, KD4_IN_ADDR, KD4_IN_DMA, KD4_IN_MASK, KD4_SGPR_IN, [q0s] "s"(q0s), [rowadv] "s"(rowadv) : KD4_CLOBBER);
            }
#undef KD4_SGPR_IN
        };

        if (nsteps > 0) issue(fetch_q0(0), 0, -1, std::false_type{});
        if (pass == 0) {
            // accumulators: a[0:127] = dV^T tuples (kh * 4 + dt), a[128:255] = dK^T tuples (8 + kh * 4 + dt); asm-owned
            acc256_zero();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (p.D < AD) {
                // head dims below 128: the V slots past D (this wave's own pieces have landed) become zeros -- dP = dO V^T must not
                // see what the Q / dO ring holds there
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (unit * 4 + i) * 4 + (lane >> 4);
                    if (((lane & 15) ^ swz_c(row)) * 8 >= p.D)
                        *(lds_u32x4a*)(uintptr_t)(lds_base + (unit * 4 + i) * 1024 + lane * 16) = u32x4a_t{0u, 0u, 0u, 0u};
                }
                wait_lgkm0();
            }
            __builtin_amdgcn_s_barrier();                  // the V tile is in LDS for every wave
            asm volatile("" ::: "memory");
        }
        for (int step = 0; step < nsteps;) {
            const int c = no_asm ? 0 : cls(step);
            if (c == 0) {
                if (step & 1) run(std::integral_constant<int, 1>{}, step);
                else run(std::integral_constant<int, 0>{}, step);
                ++step;
                continue;
            }
            int n = 1;
            while (step + n + 1 < nsteps && cls(step + n) == c) ++n;      // (the pass's last step never joins a run)
            if (step + 1 >= nsteps) n = 1;
            asm_run(c == 2, step, n);
            step += n;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last step's spare DMA has landed
    }

    // ---- sum the 4 units through LDS (fixed order), store dV then dK (x softmax scale). red[unit][kh * 64 + dt * 16 + r][lane]
    float* red = reinterpret_cast<float*>(smem);              // 4 x 128 x 64 floats = 128 KiB
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __syncthreads();                                       // every wave is out of the loop / done reading `red`
        if (which == 0) {
            acc256_to_lds<0>(red, unit, lane); acc256_to_lds<1>(red, unit, lane); acc256_to_lds<2>(red, unit, lane);
            acc256_to_lds<3>(red, unit, lane); acc256_to_lds<4>(red, unit, lane); acc256_to_lds<5>(red, unit, lane);
            acc256_to_lds<6>(red, unit, lane); acc256_to_lds<7>(red, unit, lane);
        } else {
            acc256_to_lds<8>(red, unit, lane); acc256_to_lds<9>(red, unit, lane); acc256_to_lds<10>(red, unit, lane);
            acc256_to_lds<11>(red, unit, lane); acc256_to_lds<12>(red, unit, lane); acc256_to_lds<13>(red, unit, lane);
            acc256_to_lds<14>(red, unit, lane); acc256_to_lds<15>(red, unit, lane);
        }
        __syncthreads();
        T* outp = which ? (T*)p.dK : (T*)p.dV;
        const int64_t o_sb = which ? p.dk_sb : p.dv_sb, o_st = which ? p.dk_st : p.dv_st, o_sh = which ? p.dk_sh : p.dv_sh;
        const float mul = which ? p.scale : 1.0f;
#pragma unroll
        for (int okh = 0; okh < 2; ++okh) {
            const int okey = k0 + okh * 32 + l31;
            float v[16];                                       // this wave's d tile: dt = unit
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) a += red[(u * 128 + okh * 64 + unit * 16 + r) * 64 + lane];
                v[r] = a * mul;
            }
            if (okey < T_) {
                T* op = outp + b * o_sb + (int64_t)okey * o_st + (int64_t)kvh * o_sh;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int d = unit * 32 + qd * 8 + lh * 4;
                    uint2 o;
                    o.x = pack_pair2<T>(v[4 * qd + 0], v[4 * qd + 1]);
                    o.y = pack_pair2<T>(v[4 * qd + 2], v[4 * qd + 3]);
                    if (d < p.D) *reinterpret_cast<uint2*>(op + d) = o;
                }
            }
        }
    }
}

}  // namespace

#ifdef UAMD_ATTN_TRACE
extern "C" int uamd_debug_attn_trace(unsigned* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &buf, sizeof(buf));
}
#endif

template <typename K_>
int set_lds_attr(K_ kernel, int bytes, bool* done) {
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        *done = true;
    }
    return 0;
}

extern "C" int uamd_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                             const float* LSE, void* dQ, void* dK, void* dV, float* Delta,
                             const int64_t* strides, int B, int T, int Hq, int Hk, int D, int lse_stride,
                             float scale, int causal, const int* lo, const int* hi, int dtype, void* stream) {
    if ((lo == nullptr) != (hi == nullptr)) return UAMD_ERR_ARG;
    if (B < 0 || T < 0 || Hq <= 0 || Hk <= 0) return UAMD_ERR_ARG;
    if (B == 0 || T == 0) return UAMD_OK;                     // empty batch: nothing to read (pointers may be null)
    if (!Q || !K || !V || !O || !dO || !LSE || !dQ || !dK || !dV || !Delta || !strides) return UAMD_ERR_ARG;
    if (D < 8 || D > AD || (D & 7) || Hq % Hk || lse_stride < T || (lse_stride & 31)) return UAMD_ERR_ARG;
    if (!causal && !(lo && hi)) return UAMD_ERR_ARG;          // non-causal: the (lo, hi) band of the documents is required
    const int G = Hq / Hk;
    if (G > 8) return UAMD_ERR_ARG;
    // dQ kernel: 8 waves = Gq query heads x 8 / Gq q-subtiles, Gq the largest of 1, 2, 4, 8 dividing G; the other G / Gq - 1 head
    // groups of a KV head are blocks of their own that read the same K / V head (virtual KV heads, AttnArgs::kvm)
    const int Gq = (G & 7) == 0 ? 8 : (G & 3) == 0 ? 4 : (G & 1) == 0 ? 2 : 1, kvm = G / Gq;
    for (int i = 0; i < 24; ++i)
        if (strides[i] & 7) return UAMD_ERR_ALIGN;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O) || !aligned16(dO) || !aligned16(dQ) ||
        !aligned16(dK) || !aligned16(dV) || !aligned16(LSE) || !aligned16(Delta))
        return UAMD_ERR_ALIGN;
    if (strides[1] > (1 << 22) || strides[4] > (1 << 22) || strides[7] > (1 << 22) || strides[13] > (1 << 22))
        return UAMD_ERR_ARG;
    AttnBwdArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.LSE = LSE; a.dQ = dQ; a.dK = dK; a.dV = dV; a.Delta = Delta;
    a.lo = lo; a.hi = hi;
    a.q_sb = strides[0]; a.q_st = strides[1]; a.q_sh = strides[2];
    a.k_sb = strides[3]; a.k_st = strides[4]; a.k_sh = strides[5];
    a.v_sb = strides[6]; a.v_st = strides[7]; a.v_sh = strides[8];
    a.o_sb = strides[9]; a.o_st = strides[10]; a.o_sh = strides[11];
    a.do_sb = strides[12]; a.do_st = strides[13]; a.do_sh = strides[14];
    a.dq_sb = strides[15]; a.dq_st = strides[16]; a.dq_sh = strides[17];
    a.dk_sb = strides[18]; a.dk_st = strides[19]; a.dk_sh = strides[20];
    a.dv_sb = strides[21]; a.dv_st = strides[22]; a.dv_sh = strides[23];
    a.B = B; a.T = T; a.Hq = Hq; a.Hk = Hk * kvm; a.G = Gq; a.nsub = 8 / Gq; a.kvm = kvm; a.lse_st = lse_stride; a.D = D;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    a.noncausal = causal ? 0 : 1;
    a.no_asm = (uamd_tuning_get(UAMD_TUNE_ATTN_VAR) & 4) ? 1 : 0;
    if ((int64_t)B * Hq * lse_stride * 4 >= (1ll << 31)) return UAMD_ERR_ARG;      // 32-bit lane offset between the two stat planes
    const int QT = 32 * a.nsub;
    a.nqt = (T + QT - 1) / QT;
    dim3 grid_q((unsigned)(a.nqt * a.Hk * B));
    dim3 grid_k((unsigned)(((T + KT - 1) / KT) * Hk * B));
    AttnBwdArgs ak = a;                                       // dK / dV: the real KV heads, all G query heads of each
    ak.Hk = Hk; ak.G = G; ak.kvm = 1;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int rc;
    // two launches: dQ (+ Delta, LSE log2 e) with the forward's tiling, then dK / dV with one wave per SIMD x 64 keys
    auto run = [&](auto tag) -> int {
        typedef decltype(tag) T;
        constexpr int ti = std::is_same<T, bf16_t>::value ? 0 : 1;
        static bool done_dq[2][2][3][64] = {}, done_kd[2][64] = {};
        int rc_;
        // head-dim class of the dQ kernel: 64 / 96 / 128 columns computed (attn_fwd_kernel's DC)
        auto dq = [&](auto band_c, auto dc_c) -> int {
            constexpr bool BAND_ = decltype(band_c)::value;
            constexpr int DC_ = decltype(dc_c)::value;
            int r_;
            if ((r_ = set_lds_attr(&attn_bwd_dq_kernel<T, BAND_, DC_>, ATTN_LDS, &done_dq[ti][BAND_][DC_ / 32 - 2][dev]))) return r_;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, BAND_, DC_>), grid_q, dim3(512), ATTN_LDS, st, a);
            return 0;
        };
        auto dq_b = [&](auto band_c) -> int {
            if (D > 96) return dq(band_c, std::integral_constant<int, 128>{});
            if (D > 64) return dq(band_c, std::integral_constant<int, 96>{});
            return dq(band_c, std::integral_constant<int, 64>{});
        };
        if ((rc_ = lo ? dq_b(std::true_type{}) : dq_b(std::false_type{}))) return rc_;
        if ((rc_ = uamd_launch_status())) return rc_;
        if ((rc_ = set_lds_attr(&attn_bwd_dkdv4_kernel<T>, KD4_LDS, &done_kd[ti][dev]))) return rc_;
        hipLaunchKernelGGL((attn_bwd_dkdv4_kernel<T>), grid_k, dim3(256), KD4_LDS, st, ak);
        return 0;
    };
    if (dtype == UAMD_BF16) rc = run(bf16_t{});
    else if (dtype == UAMD_F16) rc = run(f16_t{});
    else return UAMD_ERR_DTYPE;
    if (rc) return rc;
    return uamd_launch_status();
}

// {claim counter, finished workgroups} pairs of the persistent forward's dynamic deal: a pool per device, handed out round-robin
// -- launches on one stream run one after the other, and a launch leaves its pair zeroed (the last workgroup resets it), so a
// pair is shared only by launches ATTN_CTR_SLOTS apart. Allocated on first use; never inside a stream capture (nullptr then:
// the caller takes the one-block-per-item kernel).
constexpr int ATTN_CTR_SLOTS = 512;
static int* attn_ctr_slot(int dev, hipStream_t st) {
    static std::mutex mu;
    static int* pool[64] = {};
    static unsigned seq[64] = {};
    std::lock_guard<std::mutex> g(mu);
    if (!pool[dev]) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
        int* ptr = nullptr;
        if (hipMalloc((void**)&ptr, ATTN_CTR_SLOTS * 2 * sizeof(int)) != hipSuccess ||
            hipMemset(ptr, 0, ATTN_CTR_SLOTS * 2 * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        pool[dev] = ptr;
    }
    return pool[dev] + 2 * (seq[dev]++ % ATTN_CTR_SLOTS);
}

static int attn_fwd_impl(const void* Q, const void* K, const void* V, void* O, float* LSE,
                         const int64_t* strides, int B, int T, int Hq, int Hk, int D, int lse_stride,
                         float scale, int causal, const int* lo, const int* hi, int dtype, void* stream) {
    if (B < 0 || T < 0 || Hq <= 0 || Hk <= 0) return UAMD_ERR_ARG;
    if (B == 0 || T == 0) return UAMD_OK;                     // empty batch: nothing to read (pointers may be null)
    if (!Q || !K || !V || !O || !LSE || !strides) return UAMD_ERR_ARG;
    if (D < 8 || D > AD || (D & 7) || Hq % Hk || lse_stride < T) return UAMD_ERR_ARG;
    if (!causal && !(lo && hi)) return UAMD_ERR_ARG;          // non-causal: the (lo, hi) band of the documents is required
    if (causal) hi = nullptr;
    const int Gr = Hq / Hk;
    if (Gr > 8) return UAMD_ERR_ARG;
    // 8 waves = G query heads x 8 / G q-subtiles, G the largest of 1, 2, 4, 8 dividing the group size; a KV head's other head
    // groups are work items of their own over the same K / V head (virtual KV heads, AttnArgs::kvm): group sizes 3, 5, 6, 7
    const int G = (Gr & 7) == 0 ? 8 : (Gr & 3) == 0 ? 4 : (Gr & 1) == 0 ? 2 : 1, kvm = Gr / G;
    for (int i = 0; i < 12; ++i)
        if (strides[i] & 7) return UAMD_ERR_ALIGN;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return UAMD_ERR_ALIGN;
    // 32-bit per-lane byte offsets inside a 64-key tile
    if (strides[4] > (1 << 22) || strides[7] > (1 << 22)) return UAMD_ERR_ARG;
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.LSE = LSE; a.lo = lo; a.hi = hi;
    a.q_sb = strides[0]; a.q_st = strides[1]; a.q_sh = strides[2];
    a.k_sb = strides[3]; a.k_st = strides[4]; a.k_sh = strides[5];
    a.v_sb = strides[6]; a.v_st = strides[7]; a.v_sh = strides[8];
    a.o_sb = strides[9]; a.o_st = strides[10]; a.o_sh = strides[11];
    a.B = B; a.T = T; a.Hq = Hq; a.Hk = Hk * kvm; a.G = G; a.nsub = 8 / G; a.kvm = kvm; a.lse_st = lse_stride; a.D = D;
    a.scale_log2 = scale * 1.4426950408889634f;
    const int QT = 32 * a.nsub;
    a.nqt = (T + QT - 1) / QT;
    dim3 grid((unsigned)(a.nqt * a.Hk * B));
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int rc;
    static int ncu[64] = {};
    if (!ncu[dev]) {
        hipDeviceProp_t pr;
        ncu[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    // Plain causal batches with at least two work items per CU take the PERSISTENT kernel (one workgroup per CU walks the items,
    // dealt statically: +2-3 % at 4 x 2048 / 2 x 4096 tokens, profiles/r04_attn_ab_persistent.jsonl); packed / windowed batches,
    // small grids and non-causal bands take one block per item. For packed / windowed batches the persistent kernel exists in two
    // forms -- the static deal (items of one q tile differ in length by what the band cuts off: 20-50 % slower than one block per
    // item) and items CLAIMED from a counter (DYN) -- and both were measured against one block per item on eight packed / windowed
    // shapes (profiles/r06zv_attn_packed_ab.jsonl): the claimed deal recovers the imbalance (-1 .. -10 % against the static deal
    // on mixed documents) but stays 0 .. 30 % behind one block per item, because a seam of the persistent kernel costs about what
    // a block's fixed part does (~ 9 us against ~ 12) and a claim waits for a device-scope atomic (~ 3 us, memory-side). Not
    // the default; reachable for the parity tests and A/Bs: UAMD_TUNE_ATTN_VAR bit 0 = never persistent, bit 1 = always (band
    // launches: claimed items), bit 3 = band launches keep the static deal. 32-bit Q row offsets: T * q_st < 2^31.
    const int var = uamd_tuning_get(UAMD_TUNE_ATTN_VAR);
    bool persistent = !(var & 1) && D == AD && !hi && (int64_t)T * strides[1] < (1ll << 31) &&
                      ((var & 2) || (!lo && (int)grid.x >= 2 * ncu[dev]));
    a.ctr = nullptr;
    if (persistent && lo && !(var & 8)) a.ctr = attn_ctr_slot(dev, (hipStream_t)stream);    // (nullptr inside a capture: static deal)
    auto run = [&](auto tag) -> int {
        typedef decltype(tag) T;
        constexpr int ti = std::is_same<T, bf16_t>::value ? 0 : 1;
        static bool done[2][2][2][64] = {}, done_dyn[2][64] = {};
        auto go = [&](auto kernel, int nblk, int lds, bool* d) -> int {
            int r_;
            if ((r_ = set_lds_attr(kernel, lds, d))) return r_;
            hipLaunchKernelGGL(kernel, dim3(nblk), dim3(512), lds, st, a);
            return 0;
        };
        if (persistent) {
            const int nwg = (int)grid.x < ncu[dev] ? (int)grid.x : ncu[dev];
            if (a.ctr) return go(&attn_fwd_ps_kernel<T, true, true>, nwg, ATTN_PS_LDS, &done_dyn[ti][dev]);
            return lo ? go(&attn_fwd_ps_kernel<T, true, false>, nwg, ATTN_PS_LDS, &done[ti][1][1][dev])
                      : go(&attn_fwd_ps_kernel<T, false, false>, nwg, ATTN_PS_LDS, &done[ti][0][1][dev]);
        }
        static bool done_dc[2][2][2][64] = {};          // the head-dim classes 64 / 96 (128: `done` above)
        if (D > 96)
            return lo ? go(&attn_fwd_kernel<T, true, 128>, (int)grid.x, ATTN_LDS, &done[ti][1][0][dev])
                      : go(&attn_fwd_kernel<T, false, 128>, (int)grid.x, ATTN_LDS, &done[ti][0][0][dev]);
        if (D > 64)
            return lo ? go(&attn_fwd_kernel<T, true, 96>, (int)grid.x, ATTN_LDS, &done_dc[ti][1][1][dev])
                      : go(&attn_fwd_kernel<T, false, 96>, (int)grid.x, ATTN_LDS, &done_dc[ti][0][1][dev]);
        return lo ? go(&attn_fwd_kernel<T, true, 64>, (int)grid.x, ATTN_LDS, &done_dc[ti][1][0][dev])
                  : go(&attn_fwd_kernel<T, false, 64>, (int)grid.x, ATTN_LDS, &done_dc[ti][0][0][dev]);
    };
    if (dtype == UAMD_BF16) rc = run(bf16_t{});
    else if (dtype == UAMD_F16) rc = run(f16_t{});
    else return UAMD_ERR_DTYPE;
    if (rc) return rc;
    return uamd_launch_status();
}

extern "C" int uamd_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE,
                             const int64_t* strides, int B, int T, int Hq, int Hk, int D, int lse_stride,
                             float scale, int causal, const int* lo, int dtype, void* stream) {
    if (!causal) return UAMD_ERR_ARG;                         // (non-causal: uamd_attn_fwd_band)
    return attn_fwd_impl(Q, K, V, O, LSE, strides, B, T, Hq, Hk, D, lse_stride, scale, 1, lo, nullptr, dtype, stream);
}

// The same with both edges of the band: query t attends keys lo[t] .. hi[t]. causal != 0: hi is ignored (the upper edge is t).
// causal == 0: BIDIRECTIONAL attention inside documents (lo = start, hi = end of the query's document): the vision tower of
// BASELINE config 4 (Qwen2-VL's ViT attends all patches of an image / frame, `cu_seqlens` windows; the reference hands it to
// flash-attn / SDPA through the same run_attention, utils/attention_dispatch.py:298-617).
extern "C" int uamd_attn_fwd_band(const void* Q, const void* K, const void* V, void* O, float* LSE,
                                  const int64_t* strides, int B, int T, int Hq, int Hk, int D, int lse_stride,
                                  float scale, int causal, const int* lo, const int* hi, int dtype, void* stream) {
    return attn_fwd_impl(Q, K, V, O, LSE, strides, B, T, Hq, Hk, D, lse_stride, scale, causal, lo, hi, dtype, stream);
}
