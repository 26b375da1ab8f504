// LoRA gradient products for gfx950: the rank-r "TN" contractions over the token dimension.
//
// Replaces, in the backward of the reference's manual-autograd blocks (unsloth/kernels/fast_lora.py:172-189,
// :476-495, :632-637), the twelve `addmm_` / `matmul` calls per decoder layer of the form
//     d_A = s * (dY @ B)^T-ish @ X        and        d_B = s * (X @ A^T)^T @ dY
// i.e.  G[r, n] = s * sum_m P[m, r] * Z[m, n]   with P = dY·B or X·A^T ([M, r], r <= 16 per problem, fp32 from
// uamd_lora_xa, rounded to the activation dtype on load exactly where the reference holds a bf16 tensor) and
// Z = X, dY, h, df, de ... ([M, N] activations, N = 1024..14336).
//
// These are memory-bound streaming passes over Z (2 B per 2*r flops), not GEMM-shaped work for the matrix
// cores: Z is read exactly once, 8 bytes per lane per row (a wave covers a 256-column slab), the r <= 16
// coefficients of a row PAIR are wave-uniform and come from a 1 KiB LDS table by broadcast ds_read_b128, and
// the arithmetic is v_pk_fma_f32 (two columns per instruction, fp32): 32 packed FMAs + 4 unpack ops per row x 4
// columns, which keeps the VALU under the HBM time (v_dot2c_f32_bf16 was tried first: it issues at a fraction
// of the VALU rate on gfx950 and held the kernel at 2.7 TB/s). Up to 8 problems (all products of one
// autograd Function) go in ONE launch. Split over the token dimension is deterministic: every wave owns one
// (256-column slab, row chunk) unit and writes one partial slab; a second tiny kernel sums the partials in fixed
// order. No LDS reduction, no barrier: occupancy is bounded by registers only.
#include "common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

#define UAMD_TN_MAX_PROBLEMS 8

namespace {

constexpr int TN_COLS = 512;          // columns per slab (8 per lane)
constexpr int TN_R = 16;              // ranks per problem
constexpr int TN_RW = 8;              // ranks per wave

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

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// 16-bit float pair (one 32-bit word, little endian) -> two fp32
template <typename T> __device__ __forceinline__ f32x2_t unpack2(uint32_t w);
template <> __device__ __forceinline__ f32x2_t unpack2<bf16_t>(uint32_t w) {
    f32x2_t r;
    r.x = __uint_as_float(w << 16);
    r.y = __uint_as_float(w & 0xffff0000u);
    return r;
}
template <> __device__ __forceinline__ f32x2_t unpack2<f16_t>(uint32_t w) {
    union { uint32_t u; f16_t h[2]; } v;
    v.u = w;
    f32x2_t r;
    r.x = (float)v.h[0];
    r.y = (float)v.h[1];
    return r;
}

// One wave = one unit (512-column slab, 8 of the 16 ranks, row chunk). The two rank halves of a (slab, chunk) are
// adjacent waves of one block, so the second read of Z hits L1/L2. Bytes in flight are what bounds a streaming
// kernel here (~2.5 us loaded latency): 64 accumulator registers per lane leave room for 8 x 16-byte loads in
// flight per lane at 4 waves per SIMD = 128 KiB per CU.
template <typename T>
__global__ void __launch_bounds__(256, 3) lora_tn_kernel(TnArgs a) {
    __shared__ __attribute__((aligned(16))) float ptab[4 * 32 * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= (int64_t)a.total_slabs * a.S * 2) return;      // wave-uniform; no block-level sync below
    const int half = (int)(unit & 1);
    const int slab_lin = (int)((unit >> 1) % a.total_slabs);
    const int sblk = (int)((unit >> 1) / a.total_slabs);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < UAMD_TN_MAX_PROBLEMS; ++i)
        if (i < a.n_probs && slab_lin >= a.slab_start[i]) pi = i;
    const uamd_lora_tn_problem& pr = a.p[pi];
    const int slab = slab_lin - a.slab_start[pi];
    const int n_slabs = a.slab_start[pi + 1] - a.slab_start[pi];
    const int M = a.M, N = pr.N, R = pr.R;
    const int r_lo = half * TN_RW;
    const int n0 = slab * TN_COLS + lane * 8;
    // lanes past the last column read the last valid 16 bytes instead (their partials are never summed)
    const int n0c = n0 + 8 <= N ? n0 : (N >= 8 ? N - 8 : 0);
    const bool col_ok = n0 + 8 <= N;                  // else: ragged tail handled by the guarded path
    const T* Z = (const T*)pr.Z;
    const float* P = pr.P;
    float* mytab = ptab + wave * 256;                  // [32 rows][8 ranks] fp32

    f32x2_t acc[TN_RW][4];
#pragma unroll
    for (int r = 0; r < TN_RW; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = f32x2_t{0.f, 0.f};

    const bool tail_cols = (N & 7) != 0 || N < 8;     // problem-uniform: N % 4 == 0 only
    const int m_base = sblk * a.rows_per_wave;
    for (int ch = 0; ch < a.rows_per_wave / 32; ++ch) {
        const int mrow0 = m_base + ch * 32;
        if (mrow0 >= M) break;                          // wave-uniform
        // ---- stage the coefficients of these 32 rows, rounded to the activation dtype (the reference holds
        //      dY @ B / X @ A^T as tensors of that dtype): lane -> (row = lane>>1, 4 ranks at (lane&1)*4)
        {
            const int row = lane >> 1, rq = (lane & 1) * 4;
            const int gm = mrow0 + row;
            float pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r_lo + rq + i;
                const float x = (gm < M && r < R) ? P[(int64_t)gm * pr.ldp + r] : 0.f;
                pv[i] = round_to<T>(x);
            }
            *reinterpret_cast<float4*>(mytab + row * 8 + rq) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        }
        const bool full = (mrow0 + 32 <= M) && !tail_cols;          // wave-uniform
        // ---- stream the 32 rows (same wave wrote the table: LDS is in order per wave). Per row: one 16-byte
        //      load, 8 unpack ops, 32 v_pk_fma_f32 (rank coefficient broadcast to both halves).
#define TN_ROW(LOAD)                                                                                     \
        {                                                                                                \
            const uint4 z = LOAD;                                                                        \
            const f32x2_t z0 = unpack2<T>(z.x), z1 = unpack2<T>(z.y), z2 = unpack2<T>(z.z), z3 = unpack2<T>(z.w); \
            const float4 q0 = *reinterpret_cast<const float4*>(mytab + rr * 8);       /* broadcast reads */  \
            const float4 q1 = *reinterpret_cast<const float4*>(mytab + rr * 8 + 4);                       \
            const float pc[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};                         \
            _Pragma("unroll") for (int r = 0; r < TN_RW; ++r) {                                           \
                const f32x2_t pp = {pc[r], pc[r]};                                                        \
                acc[r][0] = __builtin_elementwise_fma(pp, z0, acc[r][0]);                                 \
                acc[r][1] = __builtin_elementwise_fma(pp, z1, acc[r][1]);                                 \
                acc[r][2] = __builtin_elementwise_fma(pp, z2, acc[r][2]);                                 \
                acc[r][3] = __builtin_elementwise_fma(pp, z3, acc[r][3]);                                 \
            }                                                                                             \
        }
        if (full) {
            const T* zp = Z + (int64_t)mrow0 * pr.ldz + n0c;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) TN_ROW(*reinterpret_cast<const uint4*>(zp + (int64_t)rr * pr.ldz))
        } else {
            for (int rr = 0; rr < 32; ++rr) {
                const int gm = mrow0 + rr;
                uint4 zz = make_uint4(0, 0, 0, 0);
                if (gm < M) {
                    if (col_ok) {
                        zz = *reinterpret_cast<const uint4*>(Z + (int64_t)gm * pr.ldz + n0);
                    } else if (n0 < N) {               // N % 4 == 0: exactly 4 valid columns
                        const uint2 h = *reinterpret_cast<const uint2*>(Z + (int64_t)gm * pr.ldz + n0);
                        zz.x = h.x; zz.y = h.y;
                    }
                }
                TN_ROW(zz)
            }
        }
#undef TN_ROW
    }

    // ---- partial store: part[sblk][r][n], n padded to whole slabs (2 KiB contiguous per rank per wave).
    //      On the full path, lanes past N hold sums of clamped columns: stored into padding, never reduced.
    const int64_t npad = (int64_t)n_slabs * TN_COLS;
    float* part = a.ws + a.ws_off[pi] + ((int64_t)sblk * TN_R + r_lo) * npad + slab * TN_COLS + lane * 8;
    const bool keep = tail_cols || col_ok;              // clamped lanes must not clobber real columns
    if (keep) {
#pragma unroll
        for (int r = 0; r < TN_RW; ++r) {
            *reinterpret_cast<float4*>(part + (int64_t)r * npad) =
                make_float4(acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y);
            *reinterpret_cast<float4*>(part + (int64_t)r * npad + 4) =
                make_float4(acc[r][2].x, acc[r][2].y, acc[r][3].x, acc[r][3].y);
        }
    }
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
        while (rpw > 128 && 2 * slabs_all * ((M + rpw - 1) / rpw) < 4096) rpw >>= 1;
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
            if ((p.N & 3) || (p.ldz & 3) || (reinterpret_cast<uintptr_t>(p.Z) & 7)) return UAMD_ERR_ALIGN;
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
    const int64_t blocks = ((int64_t)slabs * a.S * 2 + 3) / 4;
    if (blocks > 0x7fffffffLL) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) {
        hipLaunchKernelGGL((lora_tn_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else if (dtype == UAMD_F16) {
        hipLaunchKernelGGL((lora_tn_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        return UAMD_ERR_DTYPE;
    }
    int rc = uamd_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(lora_tn_reduce_kernel, dim3((unsigned)max_rn_blocks, (unsigned)n_probs), dim3(256), 0, st, a);
    return uamd_launch_status();
}
