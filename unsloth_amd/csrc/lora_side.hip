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
// the arithmetic is v_dot2c_f32_bf16 (two rows per instruction, fp32 accumulate): 64 dot2 + 4 v_perm per
// 2 rows x 4 columns, which keeps the VALU under the HBM time. Up to 8 problems (all products of one
// autograd Function) go in ONE launch. Split over the token dimension is deterministic: each block reduces its
// 4 waves through LDS and writes one partial slab; a second tiny kernel sums the partials in fixed order.
#include "common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

#define UAMD_TN_MAX_PROBLEMS 8

namespace {

constexpr int TN_COLS = 256;          // columns per slab (4 per lane)
constexpr int TN_ROWS_WAVE = 128;     // rows per wave
constexpr int TN_ROWS_BLOCK = 512;    // rows per block (4 waves)
constexpr int TN_R = 16;
constexpr int TN_LDS_RED = 4 * 64 * 64 * 4;          // 64 KiB: [wave][acc][lane]
constexpr int TN_LDS_P = 4 * 16 * 16 * 4;            // 4 KiB: per wave [16 pairs][16 r] packed 2x16-bit
constexpr int TN_LDS = TN_LDS_RED + TN_LDS_P;

struct TnArgs {
    int n_probs;
    int M;
    int S;                                   // number of 512-row blocks
    int total_slabs;
    int slab_start[UAMD_TN_MAX_PROBLEMS + 1];
    int64_t ws_off[UAMD_TN_MAX_PROBLEMS];    // float offset of the problem's partials in `ws`
    float* ws;
    uamd_lora_tn_problem p[UAMD_TN_MAX_PROBLEMS];
};

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

template <typename T>
__global__ void __launch_bounds__(256) lora_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                                   // [4][64][64]
    uint32_t* ptab = reinterpret_cast<uint32_t*>(smem + TN_LDS_RED);               // [4][16][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slab_lin = blockIdx.x % a.total_slabs;
    const int sblk = blockIdx.x / a.total_slabs;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < UAMD_TN_MAX_PROBLEMS; ++i)
        if (i < a.n_probs && slab_lin >= a.slab_start[i]) pi = i;
    const uamd_lora_tn_problem& pr = a.p[pi];
    const int slab = slab_lin - a.slab_start[pi];
    const int n_slabs = a.slab_start[pi + 1] - a.slab_start[pi];
    const int M = a.M, N = pr.N, R = pr.R;
    const int n0 = slab * TN_COLS + lane * 4;
    const bool col_ok = n0 < N;                         // N % 4 == 0 (host-checked)
    const T* Z = (const T*)pr.Z;
    const float* P = pr.P;
    uint32_t* mytab = ptab + wave * 256;

    float acc[TN_R][4];
#pragma unroll
    for (int r = 0; r < TN_R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;

    const int m_base = sblk * TN_ROWS_BLOCK + wave * TN_ROWS_WAVE;
    for (int ch = 0; ch < TN_ROWS_WAVE / 32; ++ch) {
        const int mrow0 = m_base + ch * 32;
        if (mrow0 >= M) break;                          // wave-uniform
        // ---- stage the coefficient pairs of these 32 rows: lane -> (pair = lane>>2, 4 ranks at (lane&3)*4)
        {
            const int pair = lane >> 2, rq = (lane & 3) * 4;
            const int rA = mrow0 + 2 * pair, rB = rA + 1;
            float pa[4], pb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pa[i] = (rA < M && rq + i < R) ? P[(int64_t)rA * pr.ldp + rq + i] : 0.f;
                pb[i] = (rB < M && rq + i < R) ? P[(int64_t)rB * pr.ldp + rq + i] : 0.f;
            }
            uint4 w;
            w.x = pack2<T>(pa[0], pb[0]); w.y = pack2<T>(pa[1], pb[1]);
            w.z = pack2<T>(pa[2], pb[2]); w.w = pack2<T>(pa[3], pb[3]);
            *reinterpret_cast<uint4*>(mytab + pair * 16 + rq) = w;
        }
        // ---- stream the 16 row pairs (same wave wrote the table: LDS is in order per wave)
#pragma unroll 4
        for (int pq = 0; pq < 16; ++pq) {
            const int rA = mrow0 + 2 * pq, rB = rA + 1;
            uint2 za = make_uint2(0, 0), zb = make_uint2(0, 0);
            if (col_ok && rA < M) za = *reinterpret_cast<const uint2*>(Z + (int64_t)rA * pr.ldz + n0);
            if (col_ok && rB < M) zb = *reinterpret_cast<const uint2*>(Z + (int64_t)rB * pr.ldz + n0);
            uint32_t q[4];
            q[0] = __builtin_amdgcn_perm(zb.x, za.x, 0x05040100u);      // (z[rA][n0+0], z[rB][n0+0])
            q[1] = __builtin_amdgcn_perm(zb.x, za.x, 0x07060302u);
            q[2] = __builtin_amdgcn_perm(zb.y, za.y, 0x05040100u);
            q[3] = __builtin_amdgcn_perm(zb.y, za.y, 0x07060302u);
            uint4 pw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[i] = *reinterpret_cast<const uint4*>(mytab + pq * 16 + i * 4);
            const uint32_t pv[16] = {pw[0].x, pw[0].y, pw[0].z, pw[0].w, pw[1].x, pw[1].y, pw[1].z, pw[1].w,
                                     pw[2].x, pw[2].y, pw[2].z, pw[2].w, pw[3].x, pw[3].y, pw[3].z, pw[3].w};
#pragma unroll
            for (int r = 0; r < TN_R; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = Dot2<T>::run(pv[r], q[c], acc[r][c]);
        }
    }

    // ---- block reduction (fixed order) and partial store: part[sblk][r][n], n padded to whole slabs
#pragma unroll
    for (int r = 0; r < TN_R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[(wave * 64 + r * 4 + c) * 64 + lane] = acc[r][c];
    __syncthreads();
    const int64_t npad = (int64_t)n_slabs * TN_COLS;
    float* part = a.ws + a.ws_off[pi] + (int64_t)sblk * TN_R * npad;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 4 + i;
        float4 o;
        float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = (r * 4 + c) * 64 + lane;
            ov[c] = ((red[e] + red[64 * 64 + e]) + red[2 * 64 * 64 + e]) + red[3 * 64 * 64 + e];
        }
        *reinterpret_cast<float4*>(part + (int64_t)r * npad + slab * TN_COLS + lane * 4) = o;
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
    a.n_probs = n_probs; a.M = M; a.S = (M + TN_ROWS_BLOCK - 1) / TN_ROWS_BLOCK; a.ws = workspace;
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
    const int64_t blocks = (int64_t)slabs * a.S;
    if (blocks > 0x7fffffffLL) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[2][64] = {{false}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (dtype == UAMD_BF16) {
        if (!attr_set[0][dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lora_tn_kernel<bf16_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set[0][dev] = true;
        }
        hipLaunchKernelGGL((lora_tn_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), TN_LDS, st, a);
    } else if (dtype == UAMD_F16) {
        if (!attr_set[1][dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lora_tn_kernel<f16_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set[1][dev] = true;
        }
        hipLaunchKernelGGL((lora_tn_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), TN_LDS, st, a);
    } else {
        return UAMD_ERR_DTYPE;
    }
    int rc = uamd_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(lora_tn_reduce_kernel, dim3((unsigned)max_rn_blocks, (unsigned)n_probs), dim3(256), 0, st, a);
    return uamd_launch_status();
}
