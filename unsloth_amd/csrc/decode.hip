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

// -DUAMD_DECODE_TRACE: s_memtime stamps at the phase boundaries of gemv_kernel and attn_decode_fused_kernel, 16 per wave
// (tools/decode_trace.py); never in the shipped library.
#ifdef UAMD_DECODE_TRACE
__device__ unsigned long long* g_dec_trace = nullptr;
#define DSTAMP(I)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        dts[(I)] = __builtin_amdgcn_s_memtime();                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define DTRACE_DECL unsigned long long dts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, __builtin_amdgcn_s_memrealtime(), 0, 0}
#define DTRACE_FLUSH(NBLK_LINEAR, WAVES)                                                           \
    do {                                                                                           \
        if (g_dec_trace && (threadIdx.x & 63) == 0) {                                              \
            unsigned long long* d = g_dec_trace + ((size_t)(NBLK_LINEAR) * (WAVES) + (threadIdx.x >> 6)) * 16; \
            dts[14] = __builtin_amdgcn_s_memrealtime();      /* 100 MHz: calibrates the s_memtime ticks of dts[0..12] */ \
            dts[15] = 1ull + (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);    /* HW_REG_XCC_ID[3:0] */ \
            for (int i_ = 0; i_ < 16; ++i_) d[i_] = dts[i_];                                       \
        }                                                                                          \
    } while (0)
#else
#define DSTAMP(I) do { } while (0)
#define DTRACE_DECL do { } while (0)
#define DTRACE_FLUSH(A, B) do { } while (0)
#endif

#define UAMD_GEMV_MAX_GROUPS 4
struct GemvArgs {
    const void* x;
    int K, n_groups, bs_shift, total_rows;         // blocksize = 1 << bs_shift
    int n_tb;                                      // leading workgroups that compute t = A x (pro.a_rows) instead of weight rows
    // what the load phase needs of each group, apart from g[] so that the kernel fetches it as ONE batch of scalar loads:
    // hA = absmax_f32 (meta bit 8 set) or absmax_u8; meta = log2(blocksize2) | direct << 8 | lora_b_f32 << 9 | R << 16 (R = 0
    // without an adapter)
    const void* hW[UAMD_GEMV_MAX_GROUPS];
    const void* hA[UAMD_GEMV_MAX_GROUPS];
    const void* hA2[UAMD_GEMV_MAX_GROUPS];
    const void* hB[UAMD_GEMV_MAX_GROUPS];
    int64_t hldw[UAMD_GEMV_MAX_GROUPS];
    int hldb[UAMD_GEMV_MAX_GROUPS];
    int hmeta[UAMD_GEMV_MAX_GROUPS];
    int hN[UAMD_GEMV_MAX_GROUPS];
    // a wave's trip = RB adjacent rows of ONE group (a group's last trip may be short), so the group is chosen once per trip:
    // group g owns trips [trip_start[g], trip_start[g + 1]). glu: a trip is RB / 2 rows n of gate and up each.
    int trip_start[UAMD_GEMV_MAX_GROUPS + 1];
    int total_trips;
    int t_ks, t_kp;                                // t = A x: K parts per row (power of 2 <= 8), columns per part (multiple of 512)
    int row_start[UAMD_GEMV_MAX_GROUPS + 1];
    uamd_gemv_group g[UAMD_GEMV_MAX_GROUPS];
    uamd_gemv_prologue pro;                        // mode 0 / a_rows NULL = the plain kernel
};

// One block = 8 waves sharing the decode table and the token x in LDS; a wave takes RB ADJACENT rows per trip and
// issues every global load of the trip (weights, absmax codes) before anything else -- on the first trip even before
// the table is built -- so 4-8 KB per wave are in flight while the fixed costs are paid. NIT = iterations over K.
//
// t = A x of the LoRA factors inside the launch (pro.a_rows): the first n_tb workgroups take no weight rows; their waves
// compute the rows of t and publish each as ONE 8-byte store {fp32 value, tag} (device scope, so it is never torn and needs
// no fence). The tag is the CALLER's: a value no earlier launch on this workspace has used (pro.tag, a host counter, plus
// UAMD_TAG_STRIDE x *pro.tag_dev for launches replayed from a hipGraph -- the decode engine's step counter), so granules of
// earlier launches never match and nothing is cleared or counted between launches. (The first version took the tag from an
// epoch word that the launch's LAST workgroup advanced: an arrival counter on one word, 512 returning atomics = 6-10 us at
// the end of every gate|up / down launch, profiles/r04x_decode_phase_trace.txt.) A weight-row wave polls the granules once,
// after the dot products of its first rows (bounded spin; the producers wait for nobody, so they always get there).
constexpr int GEMV_THREADS = 512;
// Group fields: `p.g[gi].F` per ROW with a run-time gi made every access a scalar load whose ADDRESS waited for the previous one
// -- ~20 dependent round trips (2.5 us) stood between the kernel's entry and its first weight load
// (profiles/r04u_decode_phase_trace.txt). Now a trip belongs to ONE group: its fields are fetched once per trip from the h* arrays
// (independent scalar loads at a run-time index, one wait), and kept as integers so that the loads built on them are typed
// global (gptr) rather than flat.
#define OPQ(V) asm volatile("" : "+s"(V))          /* pin a wave-uniform value in scalar registers at this point */
// a 64-bit value as a GLOBAL pointer (an integer cast to a plain pointer is a flat address: flat_load, which also counts on
// lgkmcnt and drags a wait behind every row)
template <typename V> __device__ __forceinline__ const __attribute__((address_space(1))) V* gptr(uint64_t a) {
    return (const __attribute__((address_space(1))) V*)a;
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {          // a wave-uniform 64-bit value, pinned to scalar registers
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ float bits16_to_f32(uint32_t b) {
    union { uint16_t u; T h; } v;
    v.u = (uint16_t)b;
    return to_f32(v.h);
}
template <typename T, bool NF4, int NIT, int RB>
__global__ void __launch_bounds__(GEMV_THREADS, NIT <= 16 ? 4 : 2) gemv_kernel(GemvArgs p) {     // <= 16: 2 blocks per CU
    constexpr int PAIRS = NF4 ? 16 : 4;              // 16-bit pairs of x per 16-byte weight load
    constexpr int ELEMS = 2 * PAIRS;                 // columns per lane per iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char gemv_smem[];
    // layout: x [NIT * 64 * ELEMS] T | code2 [4][256] float | lut2 [256][32] u32 (NF4 only)
    T* xs = reinterpret_cast<T*>(gemv_smem);
    float* code2 = reinterpret_cast<float*>(gemv_smem + NIT * 64 * ELEMS * sizeof(T));
    uint32_t* lut2 = reinterpret_cast<uint32_t*>(code2 + UAMD_GEMV_MAX_GROUPS * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    int K = p.K;
    OPQ(K);
    DTRACE_DECL;
    DSTAMP(0);
    {   // every 64-byte line of the ~0.9 KB kernarg segment in ONE batch of scalar loads: left to the compiler they are fetched
        // as they are needed, ~8 dependent scalar-cache misses in a row before the first weight load goes out
        const __attribute__((address_space(4))) int* ka =
            (const __attribute__((address_space(4))) int*)__builtin_amdgcn_kernarg_segment_ptr();
        int touch = 0;
#pragma unroll
        for (int o = 0; o < (int)sizeof(GemvArgs); o += 64) touch |= ka[o / 4];
        asm volatile("" ::"s"(touch));
    }
    DSTAMP(10);
    // columns past K: x is zero there, so the lane may read any valid address instead (no branches in the row loop)
    int koff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int k0 = (i * 64 + lane) * ELEMS;
        koff[i] = k0 < K ? k0 : 0;
    }
    const int n_tb = p.n_tb;
    const bool t_block = (int)blockIdx.x < n_tb;
    const int nwaves = ((int)gridDim.x - n_tb) * (GEMV_THREADS / 64);
    // the launch's tag (see above): a host constant plus, under a hipGraph, a device counter the caller advances per replay
    // (the device half is loaded unconditionally from a valid address and only combined where the tag is first needed -- behind
    //  the first rows' dot products / the t row: a load in an `if` is waited for where the `if` ends, i.e. right here)
    const int* const tag_src = p.pro.tag_dev ? p.pro.tag_dev : reinterpret_cast<const int*>(p.hW[0]);
    const unsigned tag_raw = (unsigned)__builtin_nontemporal_load(tag_src);
    const unsigned tag_mul = p.pro.tag_dev ? UAMD_TAG_STRIDE : 0u;
#define UAMD_GEMV_TAG (p.pro.tag + tag_raw * tag_mul)
    int gis[RB], ns[RB];
    float direct[RB];                                // 1: single-level fp32 absmax, 0: nested
    uint4 w[RB][NIT];
    uint32_t a8[RB][NIT];
    float a2[RB][NIT];
    uint32_t braw[RB];                               // lane j: B[n][j] of the row's adapter (raw bits), loaded with the weights
    int n_groups = p.n_groups, glu = p.pro.glu, total_trips = p.total_trips, bs_shift = p.bs_shift;
    // every kernarg scalar the load phase uses, fetched HERE and pinned: left alone, the compiler re-loads them from the kernarg
    // segment where each row needs them -- ten scalar-cache round trips in a row inside load_rows (1.6 us of the 2.9 us between
    // a wave's entry and its last load going out, profiles/r04am_decode_phase_trace.txt)
    int ts1 = p.trip_start[1], ts2 = p.trip_start[2], ts3 = p.trip_start[3];
    uint64_t gW0 = (uint64_t)p.hW[0], gW1 = (uint64_t)p.hW[1], gB0 = (uint64_t)p.hB[0], gB1 = (uint64_t)p.hB[1];
    uint64_t gA0 = (uint64_t)p.hA[0], gA1 = (uint64_t)p.hA[1], gA20 = (uint64_t)p.hA2[0], gA21 = (uint64_t)p.hA2[1];
    int gM0 = p.hmeta[0], gM1 = p.hmeta[1], gL0 = p.hldb[0], gL1 = p.hldb[1], gN0 = p.hN[0];
    int64_t gD0 = p.hldw[0], gD1 = p.hldw[1];
    OPQ(n_groups); OPQ(glu); OPQ(total_trips); OPQ(bs_shift); OPQ(ts1); OPQ(ts2); OPQ(ts3);
    OPQ(gW0); OPQ(gW1); OPQ(gB0); OPQ(gB1); OPQ(gM0); OPQ(gM1); OPQ(gL0); OPQ(gL1); OPQ(gN0);
    if (NF4) { OPQ(gA0); OPQ(gA1); OPQ(gA20); OPQ(gA21); } else { OPQ(gD0); OPQ(gD1); }
    int t_gi = 0;                                    // the current trip's group (glu: row r belongs to group r & 1)
    bool valid[RB];
    auto load_rows = [&](int trip, int r_lo, int r_hi) {      // rows [r_lo, r_hi) of the trip (compile-time bounds after inlining)
        int gi = 0;
        if (1 < n_groups && trip >= ts1) gi = 1;
        if (2 < n_groups && trip >= ts2) gi = 2;
        if (3 < n_groups && trip >= ts3) gi = 3;
        t_gi = gi;
        // the trip's group, fetched ONCE: independent scalar loads at a run-time index, one wait for all of them
        // (glu: both groups, picked by the compile-time parity of r below)
        const int n0 = glu ? trip * (RB / 2) : (trip - p.trip_start[gi]) * RB;
        const uint64_t sW = (uint64_t)p.hW[gi], sB = (uint64_t)p.hB[gi];
        const int sMeta = p.hmeta[gi], sLdb = p.hldb[gi], sN = p.hN[gi];
        uint64_t sA = 0, sA2 = 0;
        int64_t sLdw = 0;
        if constexpr (NF4) { sA = (uint64_t)p.hA[gi]; sA2 = (uint64_t)p.hA2[gi]; } else { sLdw = p.hldw[gi]; }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (r < r_lo || r >= r_hi) continue;
            const int gr = r & 1;
            const uint64_t gW = glu ? (gr ? gW1 : gW0) : sW, gB = glu ? (gr ? gB1 : gB0) : sB;
            const int meta = glu ? (gr ? gM1 : gM0) : sMeta, ldb = glu ? (gr ? gL1 : gL0) : sLdb, Ng = glu ? gN0 : sN;
            const int nraw = glu ? n0 + (r >> 1) : n0 + r;
            valid[r] = nraw < Ng;
            // (readfirstlane: the row is wave-uniform by construction; said explicitly, its 64-bit products stay on the scalar unit
            //  even where the control flow around the lane-dependent B load would otherwise pull them into vector registers)
            const int nrow = __builtin_amdgcn_readfirstlane(min(nraw, Ng - 1));   // past the group's end: its last row again, not stored
            gis[r] = glu ? gr : gi;
            ns[r] = nrow;
            const bool dir = (meta >> 8) & 1;
            direct[r] = dir ? 1.f : 0.f;
            braw[r] = 0;
            // addresses: everything that depends on the ROW is wave-uniform (scalar unit, 64 bits); a lane adds a 32-bit byte
            // offset, so every load is `global_load ..., v_off, s[base]` (14 vector instructions per load before, 64-bit)
            if constexpr (NF4) {
                const uint64_t gA = glu ? (gr ? gA1 : gA0) : sA, gA2 = glu ? (gr ? gA21 : gA20) : sA2;
                const uint64_t rowbase = uniform64((uint64_t)nrow * (uint64_t)K);   // first code of the row (K % 32 == 0)
                const uint64_t wrow = uniform64(gW + (rowbase >> 1));
                const uint64_t blk_row = rowbase >> bs_shift;                        // first absmax block the row touches
                const uint32_t rem = (uint32_t)(rowbase - (blk_row << bs_shift));    // codes of that block before the row
                const uint64_t arow = uniform64(gA + blk_row * (dir ? 4u : 1u));
                const uint32_t bs2 = meta & 0xff;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    a8[r][i] = 0;
                    const uamd_u32x4 v = __builtin_nontemporal_load(gptr<uamd_u32x4>(wrow + (uint32_t)(koff[i] >> 1)));
                    w[r][i] = make_uint4(v[0], v[1], v[2], v[3]);
                    const uint32_t dblk = (rem + (uint32_t)koff[i]) >> bs_shift;    // absmax block relative to blk_row
                    if (dir) {
                        a2[r][i] = *gptr<float>(arow + dblk * 4u);
                    } else {
                        a8[r][i] = *gptr<uint8_t>(arow + dblk);
                        // nested: the fp32 absmax-of-absmax of block (blk_row + dblk) >> bs2; the uniform part again on the base
                        const uint64_t b2row = blk_row >> bs2;
                        const uint32_t rem2 = (uint32_t)(blk_row - (b2row << bs2));
                        a2[r][i] = *gptr<float>(uniform64(gA2 + b2row * 4u) + (((rem2 + dblk) >> bs2) << 2));
                    }
                }
            } else {
                const int64_t ldw = glu ? (gr ? gD1 : gD0) : sLdw;
                const uint64_t wrow = uniform64(gW + (uint64_t)((int64_t)nrow * ldw) * sizeof(T));
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    a8[r][i] = 0;
                    a2[r][i] = 0.f;
                    const uamd_u32x4 v = *gptr<uamd_u32x4>(wrow + (uint32_t)(koff[i] * (int)sizeof(T)));
                    w[r][i] = make_uint4(v[0], v[1], v[2], v[3]);
                }
            }
            if (gB && lane < (meta >> 16)) {
                const bool f32 = (meta >> 9) & 1;
                const uint64_t brow = gB + (uint64_t)((int64_t)nrow * ldb) * (f32 ? 4u : 2u);
                if (f32) braw[r] = *gptr<uint32_t>(brow + (uint32_t)(lane * 4));
                else braw[r] = *gptr<uint16_t>(brow + (uint32_t)(lane * 2));
            }
        }
    };
    // ---- the prologue's own operands FIRST (the token / residual / norm weight vectors, the nested-absmax map): vmcnt retires
    //      loads in issue order, so behind 24 weight loads these few L2 hits waited for the whole HBM round trip and the block's
    //      set-up (staging, table, barrier) started when it should have been over (profiles/r04y_decode_phase_trace.txt)
    constexpr int VPT = (NIT * 64 * ELEMS / 8 + GEMV_THREADS - 1) / GEMV_THREADS;      // 16-byte vectors of x per thread
    const int mode = p.pro.mode;
    const T* const xp = (const T*)p.x;
    union V8 { uint4 r; T e[8]; };
    V8 pa[VPT], pb[VPT], pw[VPT];
    uint4 pwf[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    float c2v[2] = {0.f, 0.f};
    const bool pre = mode == 0 || VPT == 1;              // operands in registers (mode 1 / 2 of longer rows: the LDS version below)
    const bool use_a = pre && xp != nullptr, use_b = pre && mode != 0, use_w = pre && mode == 2 && !p.pro.w_f32;
    {
        // BRANCH-FREE: a load inside `if (mode == ..)` has to be complete where the branches join, i.e. every arm ended in an
        // s_waitcnt and the weight loads below started a memory round trip late. Every vector is loaded from a valid address
        // (the real one, or element 0 of the weights when the mode has no such operand) and zeroed afterwards if unused.
        const T* const dummy = (const T*)p.hW[0];
        const bool use_wf = VPT == 1 && mode == 2 && p.pro.w_f32;
        const T* const src_a = use_a ? xp : dummy;
        const T* const src_b = use_b ? (mode == 1 ? (const T*)p.pro.x2 : (const T*)p.pro.res) : dummy;
        const T* const src_w = use_w ? (const T*)p.pro.norm_w : dummy;
        const float* const src_wf = use_wf ? (const float*)p.pro.norm_w : (const float*)dummy;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int v = tid + k * GEMV_THREADS;
            const bool in = v * 8 < K;                                                       // K % 8 == 0 (host)
            pa[k].r = *reinterpret_cast<const uint4*>(src_a + (use_a && in ? v * 8 : 0));
            pb[k].r = *reinterpret_cast<const uint4*>(src_b + (use_b && in ? v * 8 : 0));
            pw[k].r = *reinterpret_cast<const uint4*>(src_w + (use_w && in ? v * 8 : 0));
            if (VPT == 1) {
                pwf[0] = *reinterpret_cast<const uint4*>(src_wf + (use_wf && in ? v * 8 : 0));
                pwf[1] = *reinterpret_cast<const uint4*>(src_wf + (use_wf && in ? v * 8 + 4 : 0));
            }
        }
    }
    DSTAMP(11);
    if (NF4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + q * GEMV_THREADS;             // 4 groups x 256 entries
            const int gi = i >> 8;
            const float* c2 = gi == 0 ? p.g[0].code2 : gi == 1 ? p.g[1].code2 : gi == 2 ? p.g[2].code2 : p.g[3].code2;
            if (gi < p.n_groups && c2) c2v[q] = c2[i & 255];
        }
    }
    const float nf_hi = kNF4d[(tid >> 4) & 15], nf_lo = kNF4d[tid & 15];      // (constant MEMORY: two more loads that belong up here)
    DSTAMP(12);
    int trip = t_block ? total_trips : ((int)blockIdx.x - n_tb) * (GEMV_THREADS / 64) + wave_u;
    if (trip < total_trips) load_rows(trip, 0, RB);
    // t workgroups: wave (row r, part q of KS) of t = A x streams its <= 8 vectors of the A row NOW, with everything else of the
    // launch -- A does not depend on the token -- into the registers a weight-row wave uses for its weights
    const int KS = p.t_ks, Kp = p.t_kp;                    // parts per row (1, 2, 4, 8: adjacent waves of one block), columns per part
    const int t_wi = (int)blockIdx.x * (GEMV_THREADS / 64) + wave_u;
    const int t_r = t_wi / KS, t_q = t_wi - t_r * KS;
    const bool t_wave = t_block && p.pro.a_rows && t_r < p.pro.Rt;
    uint4* const wflat = &w[0][0];
    if (t_wave) {
        const T* arow = (const T*)p.pro.a_rows + (int64_t)t_r * p.pro.ld_a;
        const int k_end = min(K, (t_q + 1) * Kp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = t_q * Kp + (j * 64 + lane) * 8;
            wflat[j] = make_uint4(0, 0, 0, 0);
            if (k0 < k_end) wflat[j] = *reinterpret_cast<const uint4*>(arow + k0);
        }
    }
    DSTAMP(1);
    // ---- block setup while those loads fly: x (zero-padded; optionally PRODUCED here: SwiGLU of two vectors, or
    //      residual add + RMSNorm), t = A x of the LoRA factors, the nested-absmax maps, the byte -> value-pair table
    float* tl = reinterpret_cast<float*>(gemv_smem + NIT * 64 * ELEMS * sizeof(T) + UAMD_GEMV_MAX_GROUPS * 256 * 4 +
                                          (NF4 ? 256 * 32 * 4 : 0));          // [64 ranks x 4 groups] + 8 + 8 reduction slots
#pragma unroll
    for (int k = 0; k < VPT; ++k) {                        // operands a mode does not have, and vectors past K: zero (the loads above
        const int v = tid + k * GEMV_THREADS;              // were unconditional; their first USE is down here, behind the weights)
        const bool in = v * 8 < K;
        if (!(use_a && in)) pa[k].r = make_uint4(0, 0, 0, 0);
        if (!(use_b && in)) pb[k].r = make_uint4(0, 0, 0, 0);
        if (!(use_w && in)) pw[k].r = make_uint4(0, 0, 0, 0);
    }
    {
        const int nvec = NIT * 64 * ELEMS / 8;
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                const int v = tid + k * GEMV_THREADS;
                if (v < nvec) reinterpret_cast<uint4*>(xs)[v] = pa[k].r;
            }
        } else if (mode == 1) {
            // x = (e * sigmoid(e)).to(T) * g: the SwiGLU of fast_swiglu_inference (llama.py:572-606), rounding points of
            // the training kernel (csrc/glu.hip); p.x = e (gate), pro.x2 = g (up)
            const T* gp = (const T*)p.pro.x2;
            for (int v = tid; v < nvec; v += GEMV_THREADS) {
                V8 a, b, o;
                o.r = make_uint4(0, 0, 0, 0);
                if (v * 8 < K) {
                    if (VPT == 1) {
                        a = pa[0];
                        b = pb[0];
                    } else {
                        a.r = *reinterpret_cast<const uint4*>(xp + v * 8);
                        b.r = *reinterpret_cast<const uint4*>(gp + v * 8);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float e = to_f32(a.e[j]);
                        const float f = e * uamd_sigmoid(e);
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
            if constexpr (VPT == 1) {            // K <= 4096 (NF4: the hidden size of every 7-8 B model): h and w stay in registers
                const int v = tid;
                V8 hv = pb[0];
                float ss = 0.f;
                if (v * 8 < K) {
                    if (xp) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) hv.e[j] = from_f32<T>(to_f32(pa[0].e[j]) + to_f32(hv.e[j]));
                    }
                    if (hp && blockIdx.x == 0) *reinterpret_cast<uint4*>(hp + v * 8) = hv.r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = to_f32(hv.e[j]); ss += f * f; }
                }
                ss = wave_sum_dpp(ss);
                if (lane == 0) tl[256 + wave_u] = ss;
                __syncthreads();
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < GEMV_THREADS / 64; ++w) tot += tl[256 + w];
                const float inv = rsqrtf(tot / (float)K + p.pro.eps);
                if (v < nvec) {
                    V8 o;
                    o.r = make_uint4(0, 0, 0, 0);
                    if (v * 8 < K) {
                        const float wfl[8] = {__uint_as_float(pwf[0].x), __uint_as_float(pwf[0].y), __uint_as_float(pwf[0].z),
                                              __uint_as_float(pwf[0].w), __uint_as_float(pwf[1].x), __uint_as_float(pwf[1].y),
                                              __uint_as_float(pwf[1].z), __uint_as_float(pwf[1].w)};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float normed = to_f32(hv.e[j]) * inv;
                            if (p.pro.w_f32) {
                                o.e[j] = from_f32<T>(normed * wfl[j]);
                            } else {                                    // (x * r).to(W.dtype) * W, product in W's dtype
                                o.e[j] = from_f32<T>(round_to<T>(round_to<T>(normed) * to_f32(pw[0].e[j])));
                            }
                        }
                    }
                    reinterpret_cast<uint4*>(xs)[v] = o.r;
                }
            } else {                             // longer rows: two passes over LDS (the register version costs an occupancy step)
                float ss = 0.f;
                for (int v = tid; v < nvec; v += GEMV_THREADS) {
                    V8 a, b;
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
                    V8 h;
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
        }
        if (NF4) {
#pragma unroll
            for (int q = 0; q < 2; ++q) code2[tid + q * GEMV_THREADS] = c2v[q];
            if (tid < 256) {       // thread e builds entry e (high nibble = even element): 32 copies = 8 x 16 bytes
                const uint32_t v = pack2<T>(nf_hi, nf_lo);
                uint4* dst = reinterpret_cast<uint4*>(lut2 + tid * 32);
#pragma unroll
                for (int c = 0; c < 8; ++c) dst[c] = make_uint4(v, v, v, v);
            }
        }
    }
    DSTAMP(2);
    __syncthreads();
    DSTAMP(3);
    // ---- t = A x for the stacked LoRA A rows of the launch's projections ([Rt, K], activation dtype), by the first n_tb
    //      workgroups: (row, K part) per wave from the vectors loaded at entry, published as {value, tag}
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(p.pro.sync);
    if (t_block && p.pro.a_rows) {
        float acc = 0.f;
        if (t_wave) {
            const int k_end = min(K, (t_q + 1) * Kp);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k0 = t_q * Kp + (j * 64 + lane) * 8;
                if (k0 < k_end) {
                    union { uint4 r; uint32_t w[4]; } av, xv;
                    av.r = wflat[j];
                    xv.r = *reinterpret_cast<const uint4*>(xs + k0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = Dot2<T>::run(av.w[q], xv.w[q], acc);
                }
            }
            acc = wave_sum_dpp(acc);
        }
        if (KS > 1) {                                   // parts of a row -> its part-0 wave, through LDS, summed in part order
            if (lane == 0) tl[256 + 8 + wave_u] = acc;
            __syncthreads();
            if (t_wave && t_q == 0) {
                acc = 0.f;
                for (int q = 0; q < KS; ++q) acc += tl[256 + 8 + wave_u + q];
            }
        }
        DSTAMP(4);
        if (t_wave && t_q == 0 && lane == 0)
            __hip_atomic_store(gran + t_r, ((unsigned long long)UAMD_GEMV_TAG << 32) | (unsigned long long)__float_as_uint(acc),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool have_t = false;
    auto fetch_t = [&]() {                            // the launch's t into LDS (every wave for itself; identical values)
        const unsigned tag = UAMD_GEMV_TAG;
        for (int i = lane; i < p.pro.Rt; i += 64) {
            unsigned long long v = 0;
            bool ok = false;
            for (int spin = 0; spin < (1 << 18); ++spin) {
                v = __hip_atomic_load(gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(v >> 32) == tag;
                if (ok) break;
                __builtin_amdgcn_s_sleep(4);
            }
            tl[i] = ok ? __uint_as_float((unsigned)v) : __builtin_nanf("");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const uint32_t* lut_lane = lut2 + (lane & 31);
    const uint4* x_lane = reinterpret_cast<const uint4*>(xs) + lane * (PAIRS / 4);

    for (bool first = true; trip < total_trips; trip += nwaves, first = false) {
        // per-trip (glu: per-parity) fields of the compute phase: scalar selects, once
        const int cgi = t_gi;
        const float sOff = p.g[cgi].offset, sScale = p.g[cgi].lora_scale;
        const int sMeta = p.hmeta[cgi];
        const int sToff = p.pro.t_off[cgi];
        const float* sLt = p.g[cgi].lora_t;
        float acc[RB];
        int n_s[RB];
        bool v_s[RB];
        const bool more = trip + nwaves < total_trips;
        // the rows in two halves: the next trip's loads go out as soon as the registers of a half are free, i.e. half of them
        // have the other half's dot products to hide behind (they used to be issued after the whole trip)
        constexpr int HB = RB >= 2 ? RB / 2 : RB;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int gr = r & 1;
            const float g_offset = glu ? p.g[gr].offset : sOff;
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
                    const float a_nested = code2[(glu ? gr : cgi) * 256 + a8[r][i]] * a2[r][i] + g_offset;
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
            const int g_meta = glu ? p.hmeta[gr] : sMeta;
            const int g_R = g_meta >> 16;
            if (g_R > 0) {
                // t from the preceding launch (lora_t) or from this launch's t workgroups (pro.a_rows: rows t_off[g] ..)
                if (p.pro.a_rows && !have_t) { DSTAMP(5); fetch_t(); have_t = true; DSTAMP(6); }
                if (lane < g_R) {
                    const int toff = glu ? p.pro.t_off[gr] : sToff;
                    const float tv = p.pro.a_rows ? tl[toff + lane] : (glu ? p.g[gr].lora_t : sLt)[lane];
                    const float bv = ((g_meta >> 9) & 1) ? __uint_as_float(braw[r]) : bits16_to_f32<T>(braw[r]);
                    acc[r] += (glu ? p.g[gr].lora_scale : sScale) * bv * tv;
                }
            }
            n_s[r] = ns[r];
            v_s[r] = valid[r];
            if (more && (r + 1) % HB == 0) load_rows(trip + nwaves, r + 1 - HB, r + 1);
        }
        float tot[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) tot[r] = wave_sum_dpp(acc[r]);
        if (first) DSTAMP(7);
        if (lane == 0) {
            if (RB >= 2 && p.pro.glu) {
                // h[n] = (e * sigmoid(e)).to(T) * g with e, g rounded to T first: what uamd_swiglu_fg makes of the two stored rows
#pragma unroll
                for (int r = 0; r + 1 < RB; r += 2) {
                    if (!v_s[r]) break;
                    float e = tot[r], gg = tot[r + 1];
                    if (p.g[0].bias) e += to_f32(((const T*)p.g[0].bias)[n_s[r]]);
                    if (p.g[1].bias) gg += to_f32(((const T*)p.g[1].bias)[n_s[r]]);
                    e = round_to<T>(e);
                    gg = round_to<T>(gg);
                    const float f = e * uamd_sigmoid(e);
                    ((T*)p.g[0].y)[n_s[r]] = from_f32<T>(round_to<T>(f) * gg);
                }
            } else {
                const void* gb = p.g[cgi].bias;
                void* gy = p.g[cgi].y;
                const int yf = p.g[cgi].y_f32;
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    if (!v_s[r]) break;
                    float v = tot[r];
                    if (gb) v += to_f32(((const T*)gb)[n_s[r]]);
                    if (yf) ((float*)gy)[n_s[r]] = v;
                    else ((T*)gy)[n_s[r]] = from_f32<T>(v);
                }
            }
        }
    }
    DSTAMP(8);
    DSTAMP(9);
    DTRACE_FLUSH(blockIdx.x, GEMV_THREADS / 64);
}

template <typename T, bool NF4, int NIT, int RB>
int launch_gemv_n(const GemvArgs& a0, hipStream_t st) {
    constexpr int ELEMS = NF4 ? 32 : 8;
    const int lds = NIT * 64 * ELEMS * (int)sizeof(T) + UAMD_GEMV_MAX_GROUPS * 256 * 4 + (NF4 ? 256 * 32 * 4 : 0) + (256 + 16) * 4;
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > 48 * 1024 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, NF4, NIT, RB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    if (a0.pro.glu && RB < 2) return UAMD_ERR_ARG;                // a wave needs the gate row and the up row in one trip
    GemvArgs a = a0;
    int trips = 0;
    for (int i = 0; i < UAMD_GEMV_MAX_GROUPS; ++i) {
        a.trip_start[i] = trips;
        if (i < a.n_groups && !a.pro.glu) trips += (a.hN[i] + RB - 1) / RB;
    }
    if (a.pro.glu) trips = (a.hN[0] + RB / 2 - 1) / (RB / 2 > 0 ? RB / 2 : 1);
    a.trip_start[UAMD_GEMV_MAX_GROUPS] = trips;
    a.total_trips = trips;
    int blocks = (trips + GEMV_THREADS / 64 - 1) / (GEMV_THREADS / 64);     // one trip per wave until the chip is full
    if (blocks > 512 - a.n_tb) blocks = 512 - a.n_tb;             // 2 blocks (16 waves) per CU, the t workgroups included: a 513th
    if (blocks < 1) blocks = 1;                                   // block starts when another one ends, i.e. costs a whole round
    hipLaunchKernelGGL((gemv_kernel<T, NF4, NIT, RB>), dim3(blocks + a.n_tb), dim3(GEMV_THREADS), lds, st, a);
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


// ---------------------------------------------------------------------------------------------------------------
// RoPE + cache append + split-KV attention + combine as ONE launch (uamd_attn_decode_fused). Same RoPE arithmetic, key
// partition and combine order as rope_append_kernel -> attn_decode_kernel -> attn_decode_combine_kernel; the keys of a
// split are accumulated chunk-wise (one running-max rescale per 8 keys of a lane group instead of one per key), so the
// result equals the three launches' to fp32 rounding, not bit for bit. What changes:
//   * every block rotates the G query heads of its KV head itself from the raw q|k|v row (G x 64 pairs; qkv is not written);
//   * the block whose split owns position len0 = kv_len[b] rotates the new k, appends k and v to the cache and takes both
//     from LDS when its loop reaches that key (a store followed by a load of the same line in one kernel would depend on the
//     vector cache's write policy);
//   * all K / V loads of a split (8 wave-loads of each for 128 keys) are issued before the first one is used: the old loop
//     paid one HBM round trip per 16 keys, eight in a row;
//   * the combine. Launches of up to 256 blocks (context 4096 at 8 KV heads): partials travel as 8-byte {value, tag}
//     granules (one device-scope store each, never torn, no fence; the caller's launch tag as in gemv_kernel) and every block
//     combines ITS 1 / nsplit of the G x 128 outputs, polling the granules of all splits in split order. Larger launches
//     (not certainly resident at once): plain partials, release fence, arrival counter; the last block of a (batch, KV head)
//     combines everything after an acquire fence -- measured at 21 us for this tail (buffer_wbl2 4.5 us, the atomic's round
//     trip 5.5 us, the cold re-read of the partials 9.5 us; profiles/r04u_decode_phase_trace.txt), the price of the two
//     launches it replaces.
struct AttnDecFusedArgs {
    const void* qkv; int64_t ld_qkv;
    const void* cos_t; const void* sin_t; int64_t ld_cs;
    const int* kv_len; const int* rope_pos;
    void* kc; void* vc; int64_t c_sb, c_sh;
    float* part; int* counters;
    void* out; int64_t o_sb;
    int Hq, Hk, s_max, nsplit, split_keys, window;
    float scale_log2;
    int gran;            // 1: partials travel as {value, tag} granules and every block combines its share (whole launch resident)
    unsigned tag;
    const int* tag_dev;
};
template <typename T, int G>
__global__ void __launch_bounds__(256) attn_decode_fused_kernel(AttnDecFusedArgs a) {
    constexpr int NI = 8;                                 // wave-loads of K (and of V) in flight: 128 keys per block trip
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 4, l16 = lane & 15;
    const int len0 = a.kv_len[b];                         // keys already in the cache = index of the new one
    const int pos = a.rope_pos ? a.rope_pos[b] : len0;
    const int len = len0 + 1;
    const int first = (a.window > 0 && len > a.window) ? len - a.window : 0;
    const int s0 = split * a.split_keys, s1 = min(s0 + a.split_keys, len);
    const bool own = len0 >= s0 && len0 < s0 + a.split_keys;
    unsigned tag = a.tag;                                   // the caller's launch tag (see gemv_kernel)
    if (a.tag_dev) tag += (unsigned)*a.tag_dev * UAMD_TAG_STRIDE;
    DTRACE_DECL;
    DSTAMP(0);
    __shared__ float q_s[G][DD];
    __shared__ __attribute__((aligned(16))) T kn[DD];
    __shared__ __attribute__((aligned(16))) T vn[DD];
    __shared__ float red[4][G][2];
    __shared__ float red_o[4][G][DD];
    __shared__ int last_s;
    const T* row = (const T*)a.qkv + (int64_t)b * a.ld_qkv;
    const T* cs = (const T*)a.cos_t + (int64_t)pos * a.ld_cs;
    const T* sn = (const T*)a.sin_t + (int64_t)pos * a.ld_cs;
    constexpr int HALF = DD / 2;
    // rounding points of rope_append_kernel (= the training kernel for 16-bit tables): every product and the sum rounded to T
    for (int i = tid; i < G * HALF; i += 256) {
        const int g = i / HALF, j = i - g * HALF;
        const T* v = row + (int64_t)(kvh * G + g) * DD;
        const float c = to_f32(cs[j]), s = to_f32(sn[j]);
        const float x1 = to_f32(v[j]), x2 = to_f32(v[j + HALF]);
        q_s[g][j] = round_to<T>(round_to<T>(x1 * c) - round_to<T>(x2 * s)) * a.scale_log2;
        q_s[g][j + HALF] = round_to<T>(round_to<T>(x2 * c) + round_to<T>(x1 * s)) * a.scale_log2;
    }
    if (own) {
        if (tid < HALF) {
            const int j = tid;
            const T* v = row + (int64_t)(a.Hq + kvh) * DD;
            const float c = to_f32(cs[j]), s = to_f32(sn[j]);
            const float x1 = to_f32(v[j]), x2 = to_f32(v[j + HALF]);
            const T r1 = from_f32<T>(round_to<T>(x1 * c) - round_to<T>(x2 * s));
            const T r2 = from_f32<T>(round_to<T>(x2 * c) + round_to<T>(x1 * s));
            kn[j] = r1;
            kn[j + HALF] = r2;
            if (len0 < a.s_max) {
                T* kd = (T*)a.kc + (int64_t)b * a.c_sb + (int64_t)kvh * a.c_sh + (int64_t)len0 * DD;
                kd[j] = r1;
                kd[j + HALF] = r2;
            }
        } else if (tid >= 64 && tid < 64 + DD) {
            const int d = tid - 64;
            const T val = row[(int64_t)(a.Hq + a.Hk + kvh) * DD + d];
            vn[d] = val;
            if (len0 < a.s_max) ((T*)a.vc)[(int64_t)b * a.c_sb + (int64_t)kvh * a.c_sh + (int64_t)len0 * DD + d] = val;
        }
    }
    DSTAMP(1);
    __syncthreads();
    DSTAMP(2);
    float m[G], l[G], o[G][8];
    float qv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { qv[g][j] = q_s[g][l16 * 8 + j]; o[g][j] = 0.f; }
    }
    const T* kb = (const T*)a.kc + (int64_t)b * a.c_sb + (int64_t)kvh * a.c_sh;
    const T* vb = (const T*)a.vc + (int64_t)b * a.c_sb + (int64_t)kvh * a.c_sh;
    const int safe = len0 > 0 ? len0 - 1 : 0;
    for (int base = max(s0, first & ~15) + wave * 4; base < s1; base += 16 * NI) {
        Vec16<T> kk[NI], vv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = base + 16 * i + grp;
            const bool cached = key < s1 && key >= first && key != len0;
            const int kl = cached ? key : safe;
            kk[i] = ld16(kb + (int64_t)kl * DD + l16 * 8);
            vv[i] = ld16(vb + (int64_t)kl * DD + l16 * 8);
        }
        DSTAMP(3);
#ifdef UAMD_DECODE_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DSTAMP(4);
#endif
        // pass 1: the scores of the chunk's keys, all (key, head) pairs independent of each other (the one-key-at-a-time online
        // softmax this replaces was a single dependent chain per head: 3.9 us for 8 keys with one wave per SIMD to hide nothing
        // behind, profiles/r04z_decode_phase_trace.txt)
        float sc[NI][G];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = base + 16 * i + grp;
            const bool valid = key < s1 && key >= first;
            if (own && key == len0) {
                kk[i] = *reinterpret_cast<const Vec16<T>*>(kn + l16 * 8);
                vv[i] = *reinterpret_cast<const Vec16<T>*>(vn + l16 * 8);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += qv[g][j] * to_f32(kk[i].e[j]);
                acc = row16_sum(acc);
                sc[i][g] = valid ? acc : -INFINITY;
            }
        }
        // pass 2: one rescale per head and chunk, then the chunk's keys accumulate independently
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mc = sc[0][g];
#pragma unroll
            for (int i = 1; i < NI; ++i) mc = fmaxf(mc, sc[i][g]);
            const float mn = fmaxf(m[g], mc);
            const float mr = mn == -INFINITY ? 0.f : mn;
            const float alpha = __builtin_amdgcn_exp2f(m[g] - mr);
            m[g] = mn;
            float ls = l[g] * alpha;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] *= alpha;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float pe = __builtin_amdgcn_exp2f(sc[i][g] - mr);
                ls += pe;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[g][j] += pe * to_f32(vv[i].e[j]);
            }
            l[g] = ls;
        }
    }
    DSTAMP(5);
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float mo = __shfl_xor(m[g], off, 64), lo = __shfl_xor(l[g], off, 64);
            const float mn = fmaxf(m[g], mo);
            const float mr = mn == -INFINITY ? 0.f : mn;
            const float aa = __builtin_amdgcn_exp2f(m[g] - mr), c = __builtin_amdgcn_exp2f(mo - mr);
            l[g] = l[g] * aa + lo * c;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * aa + __shfl_xor(o[g][j], off, 64) * c;
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
            const float aa = __builtin_amdgcn_exp2f(red[s][g][0] - mr);
            ll += red[s][g][1] * aa;
            oo += red_o[s][g][d] * aa;
        }
        if (a.gran) {       // {value, tag} granules: one 8-byte device-scope store each, no fence
            unsigned long long* pg = reinterpret_cast<unsigned long long*>(a.part) +
                                     (((int64_t)b * a.Hq + kvh * G + g) * a.nsplit + split) * (DD + 2);
            const unsigned long long hi = (unsigned long long)tag << 32;
            __hip_atomic_store(pg + d, hi | __float_as_uint(oo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(pg + DD, hi | __float_as_uint(mm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pg + DD + 1, hi | __float_as_uint(ll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            float* pp = a.part + (((int64_t)b * a.Hq + kvh * G + g) * a.nsplit + split) * (DD + 2);
            pp[d] = oo;
            if (d == 0) { pp[DD] = mm; pp[DD + 1] = ll; }
        }
    }
    if (a.gran) {
        // ---- every block of the (batch, KV head) combines ITS share of the G x 128 outputs from the granules of all splits, in
        //      split order (the old combine kernel's arithmetic). All blocks of the launch are resident (the host checks), and
        //      each has published before it polls: nobody waits for a block that has not started. Polls are bounded.
        DSTAMP(6);
        const int chunk = (G * DD + a.nsplit - 1) / a.nsplit;
        const int i_end = min((split + 1) * chunk, G * DD);
        for (int i = split * chunk + tid; i < i_end; i += 256) {
            const int g = i / DD, d = i - g * DD;
            const unsigned long long* pg = reinterpret_cast<const unsigned long long*>(a.part) +
                                           ((int64_t)b * a.Hq + kvh * G + g) * a.nsplit * (DD + 2);
            float mm = -INFINITY, ll = 0.f, oo = 0.f;
            for (int sb = 0; sb < a.nsplit; sb += 8) {
                unsigned long long gm[8], gl[8], go[8];
                for (int spin = 0; spin < (1 << 18); ++spin) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int sp = min(sb + j, a.nsplit - 1);
                        gm[j] = __hip_atomic_load(pg + sp * (DD + 2) + DD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gl[j] = __hip_atomic_load(pg + sp * (DD + 2) + DD + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        go[j] = __hip_atomic_load(pg + sp * (DD + 2) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        ok = ok && (unsigned)(gm[j] >> 32) == tag && (unsigned)(gl[j] >> 32) == tag && (unsigned)(go[j] >> 32) == tag;
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (sb + j >= a.nsplit) break;
                    const float msj = (unsigned)(gm[j] >> 32) == tag ? __uint_as_float((unsigned)gm[j]) : __builtin_nanf("");
                    const float lsj = __uint_as_float((unsigned)gl[j]), osj = __uint_as_float((unsigned)go[j]);
                    const float mn = fmaxf(mm, msj);
                    const float mr = mn == -INFINITY ? 0.f : mn;
                    const float aa = __builtin_amdgcn_exp2f(mm - mr), c = __builtin_amdgcn_exp2f(msj - mr);
                    ll = ll * aa + lsj * c;
                    oo = oo * aa + osj * c;
                    mm = mn;
                }
            }
            ((T*)a.out)[(int64_t)b * a.o_sb + (int64_t)(kvh * G + g) * DD + d] = from_f32<T>(ll > 0.f ? oo / ll : 0.f);
        }
        DSTAMP(9);
        DSTAMP(10);
        DTRACE_FLUSH((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 4);
        return;
    }
    // ---- arrival; the last block of this (batch, KV head) combines
    DSTAMP(6);
    __threadfence();
    DSTAMP(7);
    __syncthreads();
    int* cnt = a.counters + (int64_t)b * a.Hk + kvh;
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == a.nsplit - 1;
    }
    __syncthreads();
    DSTAMP(8);
    if (!last_s) {
        DTRACE_FLUSH((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 4);
        return;
    }
    __threadfence();
    DSTAMP(9);
    for (int i = tid; i < G * DD; i += 256) {
        const int g = i / DD, d = i - g * DD;
        const float* pp = a.part + ((int64_t)b * a.Hq + kvh * G + g) * a.nsplit * (DD + 2);
        float mm = -INFINITY, ll = 0.f, oo = 0.f;
        for (int sb = 0; sb < a.nsplit; sb += 4) {
            float ms[4], ls[4], os[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = min(sb + j, a.nsplit - 1);
                ms[j] = sb + j < a.nsplit ? pp[s * (DD + 2) + DD] : -INFINITY;
                ls[j] = pp[s * (DD + 2) + DD + 1];
                os[j] = pp[s * (DD + 2) + d];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mn = fmaxf(mm, ms[j]);
                const float mr = mn == -INFINITY ? 0.f : mn;
                const float aa = __builtin_amdgcn_exp2f(mm - mr), c = __builtin_amdgcn_exp2f(ms[j] - mr);
                ll = ll * aa + ls[j] * c;
                oo = oo * aa + os[j] * c;
                mm = mn;
            }
        }
        ((T*)a.out)[(int64_t)b * a.o_sb + (int64_t)(kvh * G + g) * DD + d] = from_f32<T>(ll > 0.f ? oo / ll : 0.f);
    }
    if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DSTAMP(10);
    DTRACE_FLUSH((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 4);
}

}  // namespace

#ifdef UAMD_DECODE_TRACE
extern "C" int uamd_debug_decode_trace(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dec_trace), &buf, sizeof(buf));
}
#endif

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
    if (pro && pro->a_rows && (pro->Rt <= 0 || pro->Rt > 256 || (pro->ld_a & 7) || !aligned16(pro->a_rows) || !pro->sync ||
                               ((uintptr_t)pro->sync & 7) || (pro->tag == 0 && !pro->tag_dev))) return UAMD_ERR_ARG;
    const int glu = pro ? pro->glu : 0;
    if (glu && (n_groups != 2 || groups[0].N != groups[1].N || groups[0].y_f32 || K > 8192)) return UAMD_ERR_ARG;
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
        const uamd_gemv_group& h = a.g[i];
        const bool dir = nf4 && h.absmax_f32;
        a.hW[i] = h.W;
        a.hA[i] = dir ? (const void*)h.absmax_f32 : (const void*)h.absmax_u8;
        a.hA2[i] = h.absmax2;
        a.hB[i] = h.lora_b;                              // (already NULL when the group has no LoRA term)
        a.hldw[i] = h.ldw;
        if (h.lora_b && (h.ld_lb < 0 || h.ld_lb > 0x7fffffffLL)) return UAMD_ERR_ARG;
        a.hldb[i] = (int)h.ld_lb;
        a.hN[i] = h.N;
        a.hmeta[i] = (nf4 && !dir ? h.blocksize2 : 0) | (dir ? 1 << 8 : 0) | (h.lora_b_f32 ? 1 << 9 : 0) | ((h.lora_b ? h.R : 0) << 16);
    }
    a.row_start[UAMD_GEMV_MAX_GROUPS] = rows;
    a.total_rows = rows;                              // (glu: 2 N virtual rows, gate and up rows interleaved)
    a.n_tb = 0;
    a.t_ks = 1;
    a.t_kp = 4096;
    if (pro) {
        a.pro = *pro;
        if (pro->a_rows) {                            // a wave of a t workgroup holds <= 8 vectors (4096 columns) of its A row
            int ks = 1;
            while (ks < 8 && ks * 4096 < K) ks *= 2;
            if (ks * 4096 < K) return UAMD_ERR_ARG;      // K <= 32768
            a.t_ks = ks;
            a.t_kp = ((K + ks - 1) / ks + 511) / 512 * 512;
            a.n_tb = (pro->Rt * ks + GEMV_THREADS / 64 - 1) / (GEMV_THREADS / 64);
        }
    } else {
        a.pro.mode = 0; a.pro.x2 = nullptr; a.pro.res = nullptr; a.pro.norm_w = nullptr; a.pro.h_out = nullptr;
        a.pro.a_rows = nullptr; a.pro.ld_a = 0; a.pro.Rt = 0; a.pro.w_f32 = 0; a.pro.eps = 0.f; a.pro.glu = 0;
        a.pro.sync = nullptr; a.pro.tag = 0; a.pro.tag_dev = nullptr;
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

// uamd_gemv with the token PRODUCED inside the launch (one decoder-layer step = 5 launches instead of 14):
//   pro->mode 1: x = SwiGLU(x, x2)                       (fast_swiglu_inference, llama.py:572-606: down_proj's input)
//   pro->mode 2: h = x + res (x may be NULL), x' = rmsnorm(h) * norm_w; block 0 writes h to h_out   (the residual add +
//                fast_rms_layernorm_inference in front of q|k|v, gate|up and lm_head, llama.py:1249-1364)
//   pro->a_rows: t = A x for the stacked LoRA A rows [Rt, K], computed ONCE by the launch's first ceil(Rt / 8) workgroups
//                and handed to the others through pro->sync; group g reads t[t_off[g] ..]
//                (the `mv` of fast_linear_forward, utils.py:1107-1117, without a launch of its own)
//   pro->glu:    gate | up in, h = SwiGLU out (the launch of fast_swiglu_inference's elementwise kernel disappears)
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

namespace {
// How many blocks of attn_decode_fused_kernel<T, G> are CERTAINLY resident at once on the current device: one per compute
// unit THIS device has (a CPX / SPX partition reports its own CU count), and none when a block does not fit a CU at all.
// The granule combine below polls for the other splits' partials, so a launch larger than this must not take it (a block
// that starts only after another exits would be waited for in vain: 2^18 polls, then NaN). Cached per device.
template <typename T, int G>
int attn_fused_resident_blocks() {
    static int cap[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cap[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_decode_fused_kernel<T, G>, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        cap[dev] = per_cu >= 1 && cus >= 1 ? cus : -1;
    }
    return cap[dev] > 0 ? cap[dev] : 0;
}
}  // namespace

extern "C" int uamd_attn_decode_fused(const void* qkv, int64_t ld_qkv, const void* cos_t, const void* sin_t, int64_t ld_cs,
                                      const int* kv_len, const int* rope_pos, void* k_cache, void* v_cache,
                                      int64_t cache_sb, int64_t cache_sh, float* partials, int* counters, void* out,
                                      int64_t out_sb, int B, int Hq, int Hk, int D, int s_max, int nsplit, int split_keys,
                                      int window, float scale, unsigned tag, const int* tag_dev, int dtype, void* stream) {
    if (!qkv || !cos_t || !sin_t || !kv_len || !k_cache || !v_cache || !partials || !counters || !out || B <= 0 || Hq <= 0 ||
        Hk <= 0 || Hq % Hk || s_max <= 0)
        return UAMD_ERR_ARG;
    if (D != DD || nsplit <= 0 || split_keys <= 0 || (split_keys & 15)) return UAMD_ERR_ARG;
    if (!aligned16(k_cache) || !aligned16(v_cache) || (cache_sb & 7) || (cache_sh & 7)) return UAMD_ERR_ALIGN;
    const int G = Hq / Hk;
    AttnDecFusedArgs a;
    a.qkv = qkv; a.ld_qkv = ld_qkv; a.cos_t = cos_t; a.sin_t = sin_t; a.ld_cs = ld_cs; a.kv_len = kv_len; a.rope_pos = rope_pos;
    a.kc = k_cache; a.vc = v_cache; a.c_sb = cache_sb; a.c_sh = cache_sh; a.part = partials; a.counters = counters;
    a.out = out; a.o_sb = out_sb; a.Hq = Hq; a.Hk = Hk; a.s_max = s_max; a.nsplit = nsplit; a.split_keys = split_keys;
    a.window = window; a.scale_log2 = scale * 1.4426950408889634f;
    // granule combine only when every block of the launch is certainly resident at once (one 256-thread block per CU of the
    // device this call runs on, attn_fused_resident_blocks): a block polls for the partials of its KV head's other splits
    const int64_t blocks = (int64_t)nsplit * Hk * B;
    const bool want_gran = tag != 0 || tag_dev;                                       // (no tag: the arrival-counter path)
    a.gran = 0;
    a.tag = tag;
    a.tag_dev = tag_dev;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nsplit, (unsigned)Hk, (unsigned)B);
#define UAMD_DECODE_LAUNCH(TT, GG)                                                                     \
    do {                                                                                               \
        a.gran = (want_gran && blocks <= attn_fused_resident_blocks<TT, GG>()) ? 1 : 0;                \
        hipLaunchKernelGGL((attn_decode_fused_kernel<TT, GG>), grid, dim3(256), 0, st, a);             \
    } while (0)
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
    return uamd_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Greedy next token of the decode step: argmax over the fp32 logits row (the reference: `logits.argmax(-1)` in HF's generate).
// torch's reduce kernel takes 46 us for the 128,256 logits of one token (profiles/r04z_decode_kernel_stats.csv) -- 1.6 % of the
// step for half a megabyte. Two launches: 64 blocks find (max, first index) of their slices; one block finishes. Ties go to the
// SMALLEST index (torch returns the first maximal element on this path too).
namespace {
struct MaxIdx { float v; long long i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ MaxIdx wave_best(MaxIdx m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MaxIdx t;
        t.v = __shfl_xor(m.v, o, 64);
        t.i = __shfl_xor(m.i, o, 64);
        m = better(m, t);
    }
    return m;
}
__global__ void __launch_bounds__(256) argmax_part_kernel(const float* __restrict__ x, long long n, float* __restrict__ pv,
                                                          long long* __restrict__ pi) {
    __shared__ float sv[4];
    __shared__ long long si[4];
    const float* row = x + (long long)blockIdx.y * n;
    MaxIdx m = {-INFINITY, n};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = row[i];
        if (v > m.v || (v == m.v && i < m.i) || m.i == n) { m.v = v; m.i = i; }       // (NaN never wins; all-NaN rows: index of the scan's start)
    }
    m = wave_best(m);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = m.v; si[threadIdx.x >> 6] = m.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { MaxIdx t = {sv[w], si[w]}; m = better(m, t); }
        pv[blockIdx.y * gridDim.x + blockIdx.x] = m.v;
        pi[blockIdx.y * gridDim.x + blockIdx.x] = m.i;
    }
}
__global__ void __launch_bounds__(64) argmax_final_kernel(const float* __restrict__ pv, const long long* __restrict__ pi, int parts,
                                                          long long n, long long* __restrict__ out) {
    MaxIdx m = {-INFINITY, n};
    for (int i = threadIdx.x; i < parts; i += 64) { MaxIdx t = {pv[blockIdx.x * parts + i], pi[blockIdx.x * parts + i]}; m = better(m, t); }
    m = wave_best(m);
    if (threadIdx.x == 0) out[blockIdx.x] = m.i < n ? m.i : 0;
}
}  // namespace

// out[r] = argmax_i x[r, i] over `rows` contiguous fp32 rows of n entries; workspace: rows * 64 floats + rows * 64 int64.
extern "C" int uamd_argmax_f32(const float* x, int rows, int64_t n, float* ws_val, int64_t* ws_idx, int64_t* out, void* stream) {
    if (!x || !ws_val || !ws_idx || !out || rows <= 0 || n <= 0) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(argmax_part_kernel, dim3(64, (unsigned)rows), dim3(256), 0, st, x, (long long)n, ws_val, (long long*)ws_idx);
    hipLaunchKernelGGL(argmax_final_kernel, dim3((unsigned)rows), dim3(64), 0, st, ws_val, (const long long*)ws_idx, 64, (long long)n,
                       (long long*)out);
    return uamd_launch_status();
}
