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
    int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
    int B, T, Hq, Hk, G, nsub;          // G = Hq / Hk, nsub = 8 / G q-subtiles of 32 rows per block
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

template <typename T>
__device__ __forceinline__ uint32_t pack_pair(float lo, float hi) {
    union { T h[2]; uint32_t u; } v;
    v.h[0] = from_f32<T>(lo);
    v.h[1] = from_f32<T>(hi);
    return v.u;
}

template <typename T>
__global__ void __launch_bounds__(512, 2) attn_fwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 32 * p.nsub;
    const int qtile = (int)gridDim.x - 1 - (int)blockIdx.x;          // heaviest (latest) q tiles first
    const int kvh = blockIdx.y, b = blockIdx.z;
    const int head = kvh * G + (wave % G);
    const int qs = qtile * QT + (wave / G) * 32;                      // first q position of this wave
    const int q_pos = qs + l31;
    const int q_ld = q_pos < T_ ? q_pos : T_ - 1;

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

    // ---- prologue
    issue(0, 0);
    if (nkv_blk > 1) issue(1, 1);

    const int last_tile_wave = min(qs + 31, T_ - 1) / KT;            // tiles beyond are fully masked for this wave
    for (int t = 0; t < nkv_blk; ++t) {
        if (t + 1 < nkv_blk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < nkv_blk) issue(t + 2, (t + 2) % NST);           // its stage was last read before this barrier
        if (t > last_tile_wave) continue;                            // wave-uniform: nothing to add
        const unsigned char* sk = smem + (t % NST) * STAGE_B;
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
        // ---- online softmax, log2 domain. lane: q = q_pos; register r of tile kt: key below
        const int k0 = t * KT;
        const bool need_mask = (k0 + KT - 1 > qs) || (k0 + KT > T_);
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = st[kt][r] * p.scale_log2;
                if (need_mask) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > q_pos || key >= T_) s = -INFINITY;
                }
                st[kt][r] = s;
                mt = fmaxf(mt, s);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                    // first tile: exp2(-inf) = 0
        m_run = m_new;
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[kt][r] - m_new);
                st[kt][r] = e;
                ls += e;
            }
        l_run = l_run * alpha + ls;
#pragma unroll
        for (int i = 0; i < 4; ++i) o_acc[i] *= alpha;

        // ---- O^T[d][q] += V^T P^T : per (kt, c) one k-step of 16 keys; lane half lh contracts keys
        //      32 kt + 16 c + {4 lh .. 4 lh + 3, 8 + 4 lh .. 8 + 4 lh + 3} = registers 8c .. 8c+7 of st[kt]
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                union { uint32_t w[4]; frag_t f; } pb;
#pragma unroll
                for (int j = 0; j < 4; ++j) pb.w[j] = pack_pair<T>(st[kt][8 * c + 2 * j], st[kt][8 * c + 2 * j + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int a0 = (kt * 32 + c * 16) * 256 + (v_lane ^ (dt << 6));
                    union { s16x4_t h[2]; frag_t f; } va;
                    va.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0));
                    va.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0 + 8 * 256));
                    o_acc[dt] = MfmaA<T>::run(va.f, pb.f, o_acc[dt]);
                }
            }
    }

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
        if (lh == 0) p.LSE[((int64_t)b * p.Hq + head) * T_ + q_pos] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
}

}  // namespace

extern "C" int uamd_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE,
                             const int64_t* strides, int B, int T, int Hq, int Hk, int D, float scale,
                             int causal, int dtype, void* stream) {
    if (!Q || !K || !V || !O || !LSE || !strides || B < 0 || T < 0 || Hq <= 0 || Hk <= 0) return UAMD_ERR_ARG;
    if (B == 0 || T == 0) return UAMD_OK;
    if (D != AD || !causal || Hq % Hk) return UAMD_ERR_ARG;
    const int G = Hq / Hk;
    if (G != 1 && G != 2 && G != 4 && G != 8) return UAMD_ERR_ARG;
    for (int i = 0; i < 12; ++i)
        if (strides[i] & 7) return UAMD_ERR_ALIGN;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return UAMD_ERR_ALIGN;
    // 32-bit per-lane byte offsets inside a 64-key tile
    if (strides[4] > (1 << 22) || strides[7] > (1 << 22)) return UAMD_ERR_ARG;
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.LSE = LSE;
    a.q_sb = strides[0]; a.q_st = strides[1]; a.q_sh = strides[2];
    a.k_sb = strides[3]; a.k_st = strides[4]; a.k_sh = strides[5];
    a.v_sb = strides[6]; a.v_st = strides[7]; a.v_sh = strides[8];
    a.o_sb = strides[9]; a.o_st = strides[10]; a.o_sh = strides[11];
    a.B = B; a.T = T; a.Hq = Hq; a.Hk = Hk; a.G = G; a.nsub = 8 / G;
    a.scale_log2 = scale * 1.4426950408889634f;
    const int QT = 32 * a.nsub;
    dim3 grid((unsigned)((T + QT - 1) / QT), (unsigned)Hk, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[2][64] = {{false}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (dtype == UAMD_BF16) {
        if (!attr_set[0][dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<bf16_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set[0][dev] = true;
        }
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, dim3(512), ATTN_LDS, st, a);
    } else if (dtype == UAMD_F16) {
        if (!attr_set[1][dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<f16_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set[1][dev] = true;
        }
        hipLaunchKernelGGL((attn_fwd_kernel<f16_t>), grid, dim3(512), ATTN_LDS, st, a);
    } else {
        return UAMD_ERR_DTYPE;
    }
    return uamd_launch_status();
}
