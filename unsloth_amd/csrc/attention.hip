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
#include <type_traits>

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

struct AttnArgs {
    const void* Q; const void* K; const void* V; void* O; float* LSE;
    const int* lo;                       // [B, T] first key each query may attend (NULL: 0), non-decreasing in t
    int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
    int B, T, Hq, Hk, G, nsub;          // G = Hq / Hk, nsub = 8 / G q-subtiles of 32 rows per block
    int nqt;                             // number of q tiles
    int lse_st;                          // row stride of LSE [B, Hq, lse_st] (T rounded up to 32)
    float scale_log2;                    // softmax scale * log2(e)
};

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

// BAND = false: plain causal attention, the band bookkeeping folds away at compile time (it costs ~50 VGPRs).
template <typename T, bool BAND>
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
    const int qtile = p.nqt - 1 - (int)(blockIdx.x / npairs);
    const int pair_ = (int)(blockIdx.x % npairs);
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

    // ---- Q^T operand fragments (B operand: lane -> q = l31, 8 d at 16 ks + 8 lh), kept for the whole tile loop
    frag_t qf[8];
    {
        const T* qp = (const T*)p.Q + b * p.q_sb + (int64_t)q_ld * p.q_st + (int64_t)head * p.q_sh + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            union { uint4 r; frag_t f; } u;
            u.r = *reinterpret_cast<const uint4*>(qp + ks * 16);
            qf[ks] = u.f;
        }
    }

    // ---- DMA plan: a stage = K tile (64 rows x 256 B) then V tile. One DMA instruction = 4 rows. Wave w issues
    //      pieces 2w, 2w+1 (rows 8w .. 8w+7) of K and of V. lane -> (row = 4 piece + (lane>>4), stored slot =
    //      lane & 15); the stored slot holds logical slot  s ^ (row & 15)  (K)  /  s ^ ((row & 3) << 2)  (V).
    const int nkv_blk = min((qtile * QT + QT + KT - 1) / KT, (T_ + KT - 1) / KT);      // causal: keys <= last q
    int drow[2], dks[2], dvs[2];
    unsigned koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + (lane >> 4);
        drow[i] = row;
        dks[i] = ((lane & 15) ^ (row & 15)) * 16;
        dvs[i] = ((lane & 15) ^ ((row & 3) << 2)) * 16;
        koff[i] = (unsigned)((int64_t)row * p.k_st * 2 + dks[i]);
        voff[i] = (unsigned)((int64_t)row * p.v_st * 2 + dvs[i]);
    }
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvh * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh;
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

    const int last_tile_wave = min(qs + 31, T_ - 1) / KT;            // tiles beyond are fully masked for this wave
    const int first_tile_wave = lo_w0 / KT;                          // ... and tiles before
    // One tile step; MASKED is compile-time and the tile range is split by hand (see attn_bwd_dq_kernel).
    auto step = [&](int ti, auto mode_c, bool rt_mask) {       // mode 0: no mask, 1: mask, 2: mask iff rt_mask
        constexpr int MODE = decltype(mode_c)::value;
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
            for (int ks = 0; ks < 8; ++ks) {
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
            for (int dt = 0; dt < 4; ++dt) {
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
        const int k0 = t * KT;
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = st[kt][r] * p.scale_log2;
                if (MODE == 1 || (MODE == 2 && rt_mask)) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > q_pos || key >= T_ || key < lo_q) s = -INFINITY;
                }
                st[kt][r] = s;
                mt = fmaxf(mt, s);
            }
        mt = max_across_halves(mt);
        const float m_new = fmaxf(m_run, mt);
        // a row whose band starts after this tile has seen only masked keys so far: keep the exponent finite
        const float m_ref = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_ref);                    // first tile: exp2(-inf) = 0
        m_run = m_new;
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[kt][r] - m_ref);
                st[kt][r] = e;
                ls += e;
            }
        l_run = l_run * alpha + ls;
#pragma unroll
        for (int i = 0; i < 4; ++i) o_acc[i] *= alpha;

        ASTAMP(ti, 5);
        // ---- O^T[d][q] += V^T P^T : per 16-key step u = 2 kt + c; lane half lh contracts keys
        //      16 u + {4 lh .. 4 lh + 3, 8 + 4 lh .. 8 + 4 lh + 3} = registers 8c .. 8c+7 of st[kt]
        auto pv = [&](int u, const frag_t* vsrc) {
            const int kt = u >> 1, c = u & 1;
            union { uint32_t w[4]; frag_t f; } pb;
#pragma unroll
            for (int j = 0; j < 4; ++j) pb.w[j] = pack_pair<T>(st[kt][8 * c + 2 * j], st[kt][8 * c + 2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o_acc[dt] = MfmaA<T>::run(vsrc[dt], pb.f, o_acc[dt]);
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
        const int t_diag = (qs + 1) / KT, t_rag = (T_ % KT) ? T_ / KT : nkv_blk;
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
    if (q_pos < T_) {
        T* op = (T*)p.O + b * p.o_sb + (int64_t)q_pos * p.o_st + (int64_t)head * p.o_sh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = dt * 32 + qd * 8 + lh * 4;
                uint2 o;
                o.x = pack_pair<T>(o_acc[dt][qd * 4 + 0] * inv, o_acc[dt][qd * 4 + 1] * inv);
                o.y = pack_pair<T>(o_acc[dt][qd * 4 + 2] * inv, o_acc[dt][qd * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + d) = o;
            }
        if (lh == 0) p.LSE[((int64_t)b * p.Hq + head) * p.lse_st + q_pos] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
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
    float scale, scale_log2;
};

__device__ __forceinline__ int swz_c(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <typename T, bool BAND>
__global__ void __launch_bounds__(512, 2) attn_bwd_dq_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 32 * p.nsub;
    const int npairs = p.Hk * p.B;
    const int qtile = p.nqt - 1 - (int)(blockIdx.x / npairs);
    const int pair_ = (int)(blockIdx.x % npairs);
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
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            union { uint4 r; frag_t f; T e[8]; } u, d, o;
            u.r = *reinterpret_cast<const uint4*>(qp + ks * 16);
            d.r = *reinterpret_cast<const uint4*>(dp_ + ks * 16);
            o.r = *reinterpret_cast<const uint4*>(op + ks * 16);
            qf[ks] = u.f;
            dof[ks] = d.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += to_f32(d.e[j]) * to_f32(o.e[j]);
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const int64_t stat_idx = ((int64_t)b * p.Hq + head) * p.lse_st + q_ld;
    if (lh == 0 && q_pos < T_) p.Delta[stat_idx] = delta;
    const float lse2 = p.LSE[stat_idx] * 1.4426950408889634f;
    const int lo_q = BAND ? p.lo[(int64_t)b * T_ + q_ld] : 0;
    const int lo_w0 = BAND ? __builtin_amdgcn_readfirstlane(lo_q) : 0, lo_w1 = BAND ? __builtin_amdgcn_readlane(lo_q, 31) : 0;
    const int t_first = BAND ? p.lo[(int64_t)b * T_ + min(qtile * QT, T_ - 1)] / KT : 0;

    const int nkv_blk = min((qtile * QT + QT + KT - 1) / KT, (T_ + KT - 1) / KT);
    int drow[2], dsw[2];
    unsigned koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + (lane >> 4);
        drow[i] = row;
        dsw[i] = ((lane & 15) ^ swz_c(row)) * 16;
        koff[i] = (unsigned)((int64_t)row * p.k_st * 2 + dsw[i]);
        voff[i] = (unsigned)((int64_t)row * p.v_st * 2 + dsw[i]);
    }
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvh * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 2048;
    auto issue = [&](int t, int stage) {
        const int k0 = t * KT;
        const unsigned d = dst_w + stage * STAGE_B;
        if (k0 + KT <= T_) {
            dma16x2(kbase + (int64_t)k0 * p.k_st, koff[0], koff[1], d, d + 1024);
            dma16x2(vbase + (int64_t)k0 * p.v_st, voff[0], voff[1], d + TILE_B, d + TILE_B + 1024);
        } else {
            unsigned ko[2], vo[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = min(drow[i], T_ - 1 - k0);
                ko[i] = (unsigned)((int64_t)r * p.k_st * 2 + dsw[i]);
                vo[i] = (unsigned)((int64_t)r * p.v_st * 2 + dsw[i]);
            }
            dma16x2(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], d, d + 1024);
            dma16x2(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], d + TILE_B, d + TILE_B + 1024);
        }
    };

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
    const int last_tile_wave = min(qs + 31, T_ - 1) / KT;
    const int first_tile_wave = lo_w0 / KT;
    // One tile step. MASKED is a compile-time flag and the tile range is split by hand into
    // [band-edge tiles | interior tiles | diagonal / ragged tiles]: with a run-time `need_mask` that depends on
    // the band the compiler keeps both paths' registers alive in one loop body (+50 VGPRs, spills).
    auto step = [&](int ti, auto mode_c, bool rt_mask) {       // mode 0: no mask, 1: mask, 2: mask iff rt_mask
        constexpr int MODE = decltype(mode_c)::value;
        const int t = t_first + ti;
        if (ti + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ti + 2 < nt) issue(t + 2, (ti + 2) % NST);
        if (t > last_tile_wave || t < first_tile_wave) return;
        const unsigned char* sk = smem + (ti % NST) * STAGE_B;
        const unsigned char* sv = sk + TILE_B;
        const int k0 = t * KT;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16_t st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                union { uint4 r; frag_t f; } u, w;
                u.r = *reinterpret_cast<const uint4*>(sk + kt * 32 * 256 + (r_lane ^ (ks * 32)));
                w.r = *reinterpret_cast<const uint4*>(sv + kt * 32 * 256 + (r_lane ^ (ks * 32)));
                st = MfmaA<T>::run(u.f, qf[ks], st);
                dp = MfmaA<T>::run(w.f, dof[ks], dp);
            }
            // dS^T = P^T (dP^T - Delta) * scale, masked entries 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = __builtin_amdgcn_exp2f(st[r] * p.scale_log2 - lse2);
                if (MODE == 1 || (MODE == 2 && rt_mask)) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > q_pos || key >= T_ || key < lo_q) pv = 0.f;
                }
                st[r] = pv * (dp[r] - delta) * p.scale;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                union { uint32_t w[4]; frag_t f; } sb;
#pragma unroll
                for (int j = 0; j < 4; ++j) sb.w[j] = pack_pair<T>(st[8 * c + 2 * j], st[8 * c + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
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
        const int t_diag = (qs + 1) / KT, t_rag = (T_ % KT) ? T_ / KT : nkv_blk;
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
    if (q_pos < T_) {
        T* op = (T*)p.dQ + b * p.dq_sb + (int64_t)q_pos * p.dq_st + (int64_t)head * p.dq_sh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = dt * 32 + qd * 8 + lh * 4;
                uint2 o;
                o.x = pack_pair<T>(dq_acc[dt][qd * 4 + 0], dq_acc[dt][qd * 4 + 1]);
                o.y = pack_pair<T>(dq_acc[dt][qd * 4 + 2], dq_acc[dt][qd * 4 + 3]);
                *reinterpret_cast<uint2*>(op + d) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Backward, part 2: dK, dV (KV-stationary). One block = (batch, KV head, 64 keys); wave w = (key half w&1, unit
// w>>1) where the 4 units are the G query heads of the group (G = 4), or heads x q-slices (G < 4), or two passes
// of 4 heads (G = 8). Per step every unit takes one 32-row q tile of its head:
//   S = Q K^T, dP = dO V^T  (C layout: lane = key, registers = q rows; LSE / Delta come per register quad)
//   P = exp2(S c - LSE2), dS = P (dP - Delta) scale
//   dV^T[d][key] += dO^T[d][q] P[q][key],   dK^T[d][key] += Q^T[d][q] dS[q][key]
// Q and dO tiles are staged once per step in LDS (swizzle C) and read both by rows (A operands of S, dP) and
// transposed (A operands of dV^T, dK^T); K^T lives in registers, V in LDS. The units' partial dK/dV are summed
// through LDS at the end (fixed order).
constexpr int KD_STG = 4 * 16384 + 1024;             // 4 units x (Q 8 KiB + dO 8 KiB) + stats (LSE, Delta)
constexpr int KD_V_OFF = 2 * KD_STG;                 // resident V tile
constexpr int KD_LDS = KD_V_OFF + TILE_B;            // 149,504 B

template <typename T>
__global__ void __launch_bounds__(512, 2) attn_bwd_dkdv_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int kh = wave & 1, unit = wave >> 1;
    const int hpp = G < 4 ? G : 4;                    // heads per pass
    const int npass = G / hpp, nslice = 4 / hpp;
    const int hin = unit % hpp, slice = unit / hpp;
    // key tile 0 sees every q tile (causal): heaviest first over the whole grid
    const int npairs = p.Hk * p.B;
    const int jt = (int)(blockIdx.x / npairs);
    const int pair_ = (int)(blockIdx.x % npairs);
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int k0 = jt * KT;
    const int key = k0 + kh * 32 + l31;               // this lane's key (C-layout column)
    const int key_ld = key < T_ ? key : T_ - 1;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    // band upper edge: last query that attends this lane's key (non-decreasing in key)
    const int hi_k = p.hi ? p.hi[(int64_t)b * T_ + key_ld] : T_ - 1;
    const int hi_w0 = __builtin_amdgcn_readfirstlane(hi_k), hi_w1 = __builtin_amdgcn_readlane(hi_k, 31);
    const int hi_blk = p.hi ? p.hi[(int64_t)b * T_ + min(k0 + KT - 1, T_ - 1)] : T_ - 1;

    // ---- K^T operand (lane -> key, 8 d at 16 ks + 8 lh): 8 KiB per wave. Keeping it in VGPRs next to the 128
    //      accumulator registers spills, and the LDS is full (2 x 65 KiB stages + V), so it is re-read from L2 at
    //      the top of every step into registers that are dead again after the S loop. The loads are inline asm
    //      (hipcc must not count them): issued BEFORE the step's LDS-DMA, retired by a counted vmcnt that leaves
    //      exactly the DMA in flight. The V tile stays in LDS (swizzle C, read by rows as the B operand of dP).
    const T* kp = (const T*)p.K + b * p.k_sb + (int64_t)key_ld * p.k_st + (int64_t)kvh * p.k_sh + lh * 8;
    {
        const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh + (int64_t)k0 * p.v_st;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + (lane >> 4);
            const int r = min(row, T_ - 1 - k0);
            dma16x1(vbase + (int64_t)r * p.v_st + ((lane & 15) ^ swz_c(row)) * 8,
                    lds_base + KD_V_OFF + (wave * 2 + i) * 1024);
        }
    }

    // piece i = rows 4 i + (lane>>4): source byte offset = row * stride * 2 + (dsw0 ^ ((i & 3) << 4))
    // (swz_c(row) = ((lane>>4) << 2) | (i & 3) for these rows)
    const int64_t t_st = kh ? p.do_st : p.q_st;
    const int dsw0 = ((lane & 15) ^ ((lane >> 4) << 2)) << 4;
    const unsigned trow0 = (unsigned)((int64_t)(lane >> 4) * t_st * 2);
    const unsigned tstep = (unsigned)(t_st * 8);                       // 4 rows in bytes
    const int nq32 = (T_ + 31) / 32;
    const int q32_first = k0 / 32;
    const int nsteps = (min(nq32, hi_blk / 32 + 1) - q32_first + nslice - 1) / nslice;

    // per-lane LDS read addresses inside a tile (swizzle C)
    const int r_lane = l31 * 256 + ((swz_c(l31 & 15) ^ lh) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int t_lane = (4 * lh + (sg >> 2)) * 256 +
                       ((((sg >> 2) << 2) | (((gh << 1) | ((sg >> 1) & 1)) ^ lh)) << 4) + (sg & 1) * 8;

    f32x16_t dk_acc[4], dv_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk_acc[i][r] = 0.f; dv_acc[i][r] = 0.f; }

    for (int pass = 0; pass < npass; ++pass) {
        const int head = kvh * G + pass * hpp + hin;
        const T* tbase = (kh ? (const T*)p.dO + b * p.do_sb + (int64_t)head * p.do_sh
                             : (const T*)p.Q + b * p.q_sb + (int64_t)head * p.q_sh);
        auto q0_of = [&](int step, int sl) { return (q32_first + step * nslice + sl) * 32; };
        auto issue = [&](int step, int stage) {
            int q0 = q0_of(step, slice);
            if (q0 >= T_) q0 = (nq32 - 1) * 32;                  // idle unit this step: any valid tile
            const unsigned d = lds_base + stage * KD_STG + unit * 16384 + kh * 8192;
            unsigned o[8];
            if (q0 + 32 <= T_) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = trow0 + i * tstep + (unsigned)(dsw0 ^ ((i & 3) << 4));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = min(i * 4 + (lane >> 4), T_ - 1 - q0);
                    o[i] = (unsigned)((int64_t)r * t_st * 2) + (unsigned)(dsw0 ^ ((i & 3) << 4));
                }
            }
            dma16x4g(tbase + (int64_t)q0 * t_st, o[0], o[1], o[2], o[3], d);
            dma16x4g(tbase + (int64_t)q0 * t_st, o[4], o[5], o[6], o[7], d + 4096);
            {
                // stats, 1 KiB: lanes 0-31 LSE, 32-63 Delta; unit (lane>>3)&3, 4 floats at q0_u + 4 (lane&7). Every
                // wave issues the same copy, so all waves count 9 DMA instructions per step.
                const int su = (lane >> 3) & 3;
                int sq0 = q0_of(step, su / hpp);
                if (sq0 >= T_) sq0 = (nq32 - 1) * 32;
                const int sh = kvh * G + pass * hpp + (su % hpp);
                const float* sp = (lane < 32 ? p.LSE : p.Delta) + ((int64_t)b * p.Hq + sh) * p.lse_st + sq0 + 4 * (lane & 7);
                dma16x1(sp, lds_base + stage * KD_STG + 65536);
            }
        };

        issue(0, 0);
        for (int step = 0; step < nsteps; ++step) {
            const int stage = step & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // K operand loads first (older than the DMA below), unconditionally: no phi, no compiler copies of
            // registers whose data has not landed
            uamd_u32x4 kr0, kr1, kr2, kr3, kr4, kr5, kr6, kr7;
            asm volatile(
                "global_load_dwordx4 %0, %8, off\n\t"
                "global_load_dwordx4 %1, %8, off offset:32\n\t"
                "global_load_dwordx4 %2, %8, off offset:64\n\t"
                "global_load_dwordx4 %3, %8, off offset:96\n\t"
                "global_load_dwordx4 %4, %8, off offset:128\n\t"
                "global_load_dwordx4 %5, %8, off offset:160\n\t"
                "global_load_dwordx4 %6, %8, off offset:192\n\t"
                "global_load_dwordx4 %7, %8, off offset:224"
                : "=&v"(kr0), "=&v"(kr1), "=&v"(kr2), "=&v"(kr3), "=&v"(kr4), "=&v"(kr5), "=&v"(kr6), "=&v"(kr7)
                : "v"(kp)
                : "memory");
            const bool more = step + 1 < nsteps;
            if (more) issue(step + 1, stage ^ 1);                          // 9 DMA instructions per wave
            if (more) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // K landed, DMA still in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const int q0 = q0_of(step, slice);
            if (q0 >= T_ || q0 + 31 < k0 + kh * 32 || q0 > hi_w1) continue;   // idle / above the diagonal / below the band
            const unsigned char* sq = smem + stage * KD_STG + unit * 16384;
            const unsigned char* sdo = sq + 8192;
            const unsigned char* sv = smem + KD_V_OFF + kh * 32 * 256;
            const float* stats = reinterpret_cast<const float*>(smem + stage * KD_STG + 65536) + unit * 32;

            f32x16_t sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                union { uint4 r; frag_t f; } qa;
                union { uamd_u32x4 r; frag_t f; } kb;
                qa.r = *reinterpret_cast<const uint4*>(sq + (r_lane ^ (ks * 32)));
                kb.r = ks == 0 ? kr0 : ks == 1 ? kr1 : ks == 2 ? kr2 : ks == 3 ? kr3 : ks == 4 ? kr4 : ks == 5 ? kr5
                                                                                               : ks == 6 ? kr6 : kr7;
                sc = MfmaA<T>::run(qa.f, kb.f, sc);
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                union { uint4 r; frag_t f; } da, vb;
                da.r = *reinterpret_cast<const uint4*>(sdo + (r_lane ^ (ks * 32)));
                vb.r = *reinterpret_cast<const uint4*>(sv + (r_lane ^ (ks * 32)));
                dp = MfmaA<T>::run(da.f, vb.f, dp);
            }
            const bool need_mask = (q0 < k0 + kh * 32 + 31) || (q0 + 32 > T_) || (k0 + KT > T_) || (q0 + 31 > hi_w0);
            auto soft = [&](auto masked) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float4 l4 = *reinterpret_cast<const float4*>(stats + 8 * a + 4 * lh);
                    const float4 d4 = *reinterpret_cast<const float4*>(stats + 128 + 8 * a + 4 * lh);
                    const float lvv[4] = {l4.x, l4.y, l4.z, l4.w}, dlv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * a + j;
                        const float lv = lvv[j], dl = dlv[j];
                        float pv = __builtin_amdgcn_exp2f(sc[r] * p.scale_log2 - lv * 1.4426950408889634f);
                        float ds = pv * (dp[r] - dl) * p.scale;
                        if (decltype(masked)::value) {
                            const int q = q0 + 8 * a + 4 * lh + j;
                            if (key > q || q >= T_ || key >= T_ || q > hi_k) { pv = 0.f; ds = 0.f; }
                        }
                        sc[r] = pv;
                        dp[r] = ds;
                    }
                }
            };
            if (need_mask) soft(std::true_type{}); else soft(std::false_type{});
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                union { uint32_t w[4]; frag_t f; } pb, sb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pb.w[j] = pack_pair<T>(sc[8 * c + 2 * j], sc[8 * c + 2 * j + 1]);
                    sb.w[j] = pack_pair<T>(dp[8 * c + 2 * j], dp[8 * c + 2 * j + 1]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int a0 = (c * 16) * 256 + (t_lane ^ (dt << 6));
                    union { s16x4_t h[2]; frag_t f; } ta, tq;
                    ta.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sdo + a0));
                    ta.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sdo + (a0 ^ 32) + 8 * 256));
                    tq.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sq + a0));
                    tq.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sq + (a0 ^ 32) + 8 * 256));
                    dv_acc[dt] = MfmaA<T>::run(ta.f, pb.f, dv_acc[dt]);
                    dk_acc[dt] = MfmaA<T>::run(tq.f, sb.f, dk_acc[dt]);
                }
            }
        }
        __builtin_amdgcn_s_barrier();          // all reads of the last stages done before the next pass / reduction
    }

    // ---- sum the 4 units per key half through LDS (fixed order), store dV then dK
    float* red = reinterpret_cast<float*>(smem);              // [8 waves][64 regs][64 lanes] = 128 KiB
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which) __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wave * 64 + dt * 16 + r) * 64 + lane] = which ? dk_acc[dt][r] : dv_acc[dt][r];
        __syncthreads();
        T* outp = which ? (T*)p.dK : (T*)p.dV;
        const int64_t o_sb = which ? p.dk_sb : p.dv_sb, o_st = which ? p.dk_st : p.dv_st, o_sh = which ? p.dk_sh : p.dv_sh;
#pragma unroll
        for (int okh = 0; okh < 2; ++okh) {
            const int okey = k0 + okh * 32 + l31;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = wave * 8 + j;
                float a = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) a += red[((2 * u + okh) * 64 + r) * 64 + lane];
                v[j] = a;
            }
            if (okey < T_) {
                T* op = outp + b * o_sb + (int64_t)okey * o_st + (int64_t)kvh * o_sh;
                const int dt = wave >> 1, qd0 = 2 * (wave & 1);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int d = dt * 32 + (qd0 + h2) * 8 + lh * 4;
                    uint2 o;
                    o.x = pack_pair<T>(v[4 * h2 + 0], v[4 * h2 + 1]);
                    o.y = pack_pair<T>(v[4 * h2 + 2], v[4 * h2 + 3]);
                    *reinterpret_cast<uint2*>(op + d) = o;
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
    if (D != AD || !causal || Hq % Hk || lse_stride < T || (lse_stride & 31)) return UAMD_ERR_ARG;
    const int G = Hq / Hk;
    if (G != 1 && G != 2 && G != 4 && G != 8) return UAMD_ERR_ARG;
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
    a.B = B; a.T = T; a.Hq = Hq; a.Hk = Hk; a.G = G; a.nsub = 8 / G; a.lse_st = lse_stride;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    const int QT = 32 * a.nsub;
    a.nqt = (T + QT - 1) / QT;
    dim3 grid_q((unsigned)(a.nqt * Hk * B));
    dim3 grid_k((unsigned)(((T + KT - 1) / KT) * Hk * B));
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[6][64] = {{false}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int rc;
    if (dtype == UAMD_BF16) {
        if ((rc = set_lds_attr(&attn_bwd_dkdv_kernel<bf16_t>, KD_LDS, &attr_set[1][dev]))) return rc;
        if (lo) {
            if ((rc = set_lds_attr(&attn_bwd_dq_kernel<bf16_t, true>, ATTN_LDS, &attr_set[0][dev]))) return rc;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t, true>), grid_q, dim3(512), ATTN_LDS, st, a);
        } else {
            if ((rc = set_lds_attr(&attn_bwd_dq_kernel<bf16_t, false>, ATTN_LDS, &attr_set[4][dev]))) return rc;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t, false>), grid_q, dim3(512), ATTN_LDS, st, a);
        }
        if ((rc = uamd_launch_status())) return rc;
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<bf16_t>), grid_k, dim3(512), KD_LDS, st, a);
    } else if (dtype == UAMD_F16) {
        if ((rc = set_lds_attr(&attn_bwd_dkdv_kernel<f16_t>, KD_LDS, &attr_set[3][dev]))) return rc;
        if (lo) {
            if ((rc = set_lds_attr(&attn_bwd_dq_kernel<f16_t, true>, ATTN_LDS, &attr_set[2][dev]))) return rc;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<f16_t, true>), grid_q, dim3(512), ATTN_LDS, st, a);
        } else {
            if ((rc = set_lds_attr(&attn_bwd_dq_kernel<f16_t, false>, ATTN_LDS, &attr_set[5][dev]))) return rc;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<f16_t, false>), grid_q, dim3(512), ATTN_LDS, st, a);
        }
        if ((rc = uamd_launch_status())) return rc;
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<f16_t>), grid_k, dim3(512), KD_LDS, st, a);
    } else {
        return UAMD_ERR_DTYPE;
    }
    return uamd_launch_status();
}

extern "C" int uamd_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE,
                             const int64_t* strides, int B, int T, int Hq, int Hk, int D, int lse_stride,
                             float scale, int causal, const int* lo, int dtype, void* stream) {
    if (B < 0 || T < 0 || Hq <= 0 || Hk <= 0) return UAMD_ERR_ARG;
    if (B == 0 || T == 0) return UAMD_OK;                     // empty batch: nothing to read (pointers may be null)
    if (!Q || !K || !V || !O || !LSE || !strides) return UAMD_ERR_ARG;
    if (D != AD || !causal || Hq % Hk || lse_stride < T) return UAMD_ERR_ARG;
    const int G = Hq / Hk;
    if (G != 1 && G != 2 && G != 4 && G != 8) return UAMD_ERR_ARG;
    for (int i = 0; i < 12; ++i)
        if (strides[i] & 7) return UAMD_ERR_ALIGN;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return UAMD_ERR_ALIGN;
    // 32-bit per-lane byte offsets inside a 64-key tile
    if (strides[4] > (1 << 22) || strides[7] > (1 << 22)) return UAMD_ERR_ARG;
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.LSE = LSE; a.lo = lo;
    a.q_sb = strides[0]; a.q_st = strides[1]; a.q_sh = strides[2];
    a.k_sb = strides[3]; a.k_st = strides[4]; a.k_sh = strides[5];
    a.v_sb = strides[6]; a.v_st = strides[7]; a.v_sh = strides[8];
    a.o_sb = strides[9]; a.o_st = strides[10]; a.o_sh = strides[11];
    a.B = B; a.T = T; a.Hq = Hq; a.Hk = Hk; a.G = G; a.nsub = 8 / G; a.lse_st = lse_stride;
    a.scale_log2 = scale * 1.4426950408889634f;
    const int QT = 32 * a.nsub;
    a.nqt = (T + QT - 1) / QT;
    dim3 grid((unsigned)(a.nqt * Hk * B));
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[4][64] = {{false}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int rc;
    if (dtype == UAMD_BF16) {
        if (lo) {
            if ((rc = set_lds_attr(&attn_fwd_kernel<bf16_t, true>, ATTN_LDS, &attr_set[0][dev]))) return rc;
            hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, true>), grid, dim3(512), ATTN_LDS, st, a);
        } else {
            if ((rc = set_lds_attr(&attn_fwd_kernel<bf16_t, false>, ATTN_LDS, &attr_set[1][dev]))) return rc;
            hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, false>), grid, dim3(512), ATTN_LDS, st, a);
        }
    } else if (dtype == UAMD_F16) {
        if (lo) {
            if ((rc = set_lds_attr(&attn_fwd_kernel<f16_t, true>, ATTN_LDS, &attr_set[2][dev]))) return rc;
            hipLaunchKernelGGL((attn_fwd_kernel<f16_t, true>), grid, dim3(512), ATTN_LDS, st, a);
        } else {
            if ((rc = set_lds_attr(&attn_fwd_kernel<f16_t, false>, ATTN_LDS, &attr_set[3][dev]))) return rc;
            hipLaunchKernelGGL((attn_fwd_kernel<f16_t, false>), grid, dim3(512), ATTN_LDS, st, a);
        }
    } else {
        return UAMD_ERR_DTYPE;
    }
    return uamd_launch_status();
}
