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
constexpr int TN_WAVE_LDS = 2 * TN_STAGE;        // 20 KiB per wave, 80 KiB per block

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
    if (nch > 0) issue(m_base, 0);
    for (int ch = 0; ch < nch; ++ch) {
        const int m0 = m_base + ch * 32;
        const int stage = ch & 1;
        if (ch + 1 < nch) {
            issue(m0 + 32, stage ^ 1);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    float v = 0.f;
    for (int s = 0; s < a.S; ++s) v += part[(int64_t)s * TN_R * npad];
    v *= pr.scale;
    if (pr.out_nr) pr.out[(int64_t)n * pr.ldo + r] = v;
    else pr.out[(int64_t)r * pr.ldo + n] = v;
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
        int rpw = 512;
        while (rpw > 128 && slabs_all * ((M + rpw - 1) / rpw) < 4096) rpw >>= 1;
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
