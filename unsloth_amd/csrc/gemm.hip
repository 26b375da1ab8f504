// MFMA GEMMs of the LoRA / QLoRA linear layers for gfx950.
//
// Replaces, on the reference's hot path:
//   unsloth/kernels/utils.py:1128-1170  matmul_lora: out = X @ dequant(W).T ; out += (X @ A.T) @ (s * B.T)
//   unsloth/kernels/fast_lora.py:156,193-204,497-517,639-647  the dX GEMMs of the backward
// where the reference issues: 2 bitsandbytes dequant launches + `+= offset` + a cuBLAS/rocBLAS
// GEMM on a 2 B/param scratch copy of W + 2 small LoRA GEMMs per projection.
//
// Kernels here:
//   * gemm_nt_kernel<T, NF4>: C[M,N] (+)= A[M,K] @ B[N,K]^T with fp32 MFMA accumulation
//       - NF4=false: B is a dense bf16/fp16 [N,K] matrix (e.g. lm_head, or W^T produced by the
//         transposing dequant for the backward contraction over `out`)
//       - NF4=true : B is the bitsandbytes-format packed NF4 weight; nibbles are decoded and
//         scaled while the tile is staged into LDS, so the bf16 copy of W never exists in HBM
//       - optional LoRA term folded into the same accumulators BEFORE the main K loop:
//         acc = s * (bf16(XA) @ LB^T), then acc += A @ B^T (one rounding at the end)
//       - up to 3 "groups" (q/k/v or gate/up) share A and one launch, so small-N projections
//         (k/v: N=1024) still fill 256 CUs
//   * lora_xa_kernel<T>: XA[M,R] = X[M,K] @ A[R,K]^T, skinny N (R = sum of LoRA ranks of the
//     projections sharing X), fp32 out, deterministic 4-way split-K inside the block.
//
// CDNA4 mapping: 256-thread block = 4 waves (2x2), 128x128x64 tile, each wave 64x64 =
// 4x4 v_mfma_f32_16x16x32 tiles (64 fp32 accumulators / lane). Operands are staged
// global -> registers -> LDS (issue loads for tile t+1, compute tile t, then write: the HBM
// latency hides under 32 MFMAs per wave), LDS double-buffered, ONE barrier per K step.
// LDS rows are 128 B; the 16-byte slot index is XOR-swizzled with (row>>1)&7 so the
// ds_read_b128 fragment reads (16 rows x 4 k-slots per 16-lane group) are bank-conflict free.
// MFMA operands are passed swapped (W rows as the A operand) so each lane ends up with 4
// consecutive output columns -> 8-byte stores.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define UAMD_GEMM_MAX_GROUPS 3

namespace {

struct GemmArgs {
    const void* A;
    int64_t lda;
    int M, K;
    int n_groups;
    int accumulate;
    int tiles_m;
    int tile_start[UAMD_GEMM_MAX_GROUPS + 1];   // prefix sum of n-tiles per group
    uamd_gemm_group g[UAMD_GEMM_MAX_GROUPS];
};

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<f16_t> {
    typedef f16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <typename T>
__device__ __forceinline__ typename Mfma<T>::frag as_frag(uint4 raw) {
    union { uint4 r; typename Mfma<T>::frag f; } u;
    u.r = raw;
    return u.f;
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage
constexpr int SMEM_BYTES = 4 * TILE_BYTES;       // A,B x 2 stages = 64 KiB

__device__ __forceinline__ int swz(int row, int slot) {   // byte offset inside a [128][64] 16-bit tile
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

__constant__ float kNF4g[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

template <typename T, bool NF4>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Mfma<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- tile -> (group, m0, n0). m fastest so co-scheduled blocks share a B panel.
    const int tile = blockIdx.x;
    const int tn_lin = tile / p.tiles_m;
    const int tm = tile - tn_lin * p.tiles_m;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < UAMD_GEMM_MAX_GROUPS; ++i)
        if (i < p.n_groups && tn_lin >= p.tile_start[i]) gi = i;
    const uamd_gemm_group& g = p.g[gi];
    const int m0 = tm * BM, n0 = (tn_lin - p.tile_start[gi]) * BN;
    const int M = p.M, K = p.K, N = g.N;

    float* lut = reinterpret_cast<float*>(smem + SMEM_BYTES);   // NF4 only (16 floats after the tiles)
    if (NF4) {
        if (tid < 16) lut[tid] = kNF4g[tid];
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- LoRA term first: acc = s * (T(XA) @ LB^T)   (utils.py:1162-1168)
    if (g.lora_xa != nullptr) {
        const int R = g.R;
        for (int k0 = 0; k0 < R; k0 += 32) {
            const int k = k0 + l4 * 8;
            frag_t xa[4], lb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 64 + i * 16 + l15;
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
                xa[i] = as_frag<T>(v.raw);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + l15;
                uint4 raw = make_uint4(0, 0, 0, 0);
                if (n < N && k < R)
                    raw = *reinterpret_cast<const uint4*>((const T*)g.lora_b + (int64_t)n * g.ld_lb + k);
                lb[j] = as_frag<T>(raw);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<T>::run(lb[j], xa[i], acc[i][j]);
        }
        const float s = g.lora_scale;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= s;
    }

    // ---- main loop: register-staged, LDS double-buffered
    const int nk = (K + BK - 1) / BK;
    uint4 ra[4], rb[4];
    float rb_absmax = 0.f;
    const T* Ag = (const T*)p.A;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, kc = c & 7;
            const int m = m0 + row, k = k0 + kc * 8;
            ra[i] = (m < M && k < K) ? *reinterpret_cast<const uint4*>(Ag + (int64_t)m * p.lda + k)
                                     : make_uint4(0, 0, 0, 0);
        }
        if (!NF4) {
            const T* Bg = (const T*)g.B;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * 256;
                const int row = c >> 3, kc = c & 7;
                const int n = n0 + row, k = k0 + kc * 8;
                rb[i] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(Bg + (int64_t)n * g.ldb + k)
                                         : make_uint4(0, 0, 0, 0);
            }
        } else {
            // 128 rows x 64 k = 4096 packed bytes: thread -> (row = tid>>1, 32 consecutive k)
            const int row = tid >> 1, half = tid & 1;
            const int n = n0 + row;
            if (n < N) {
                const int64_t e0 = (int64_t)n * K + k0 + half * 32;      // K % 64 == 0 (host-checked)
                rb[0] = *reinterpret_cast<const uint4*>((const uint8_t*)g.B + (e0 >> 1));
                rb_absmax = g.absmax[e0 >> 6];
            } else {
                rb[0] = make_uint4(0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u);  // code 7 = 0.0
                rb_absmax = 0.f;
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * 2 * TILE_BYTES;
        unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, kc = c & 7;
            *reinterpret_cast<uint4*>(sa + swz(row, kc)) = ra[i];
        }
        if (!NF4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * 256;
                const int row = c >> 3, kc = c & 7;
                *reinterpret_cast<uint4*>(sb + swz(row, kc)) = rb[i];
            }
        } else {
            const int row = tid >> 1, half = tid & 1;
            const uint32_t w4[4] = {rb[0].x, rb[0].y, rb[0].z, rb[0].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // 4 bytes -> 8 elements -> one 16-byte slot
                Vec16<T> o;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t byte = (w4[q] >> (8 * b)) & 0xff;
                    o.e[2 * b] = from_f32<T>(lut[byte >> 4] * rb_absmax);       // high nibble = even element
                    o.e[2 * b + 1] = from_f32<T>(lut[byte & 15] * rb_absmax);
                }
                *reinterpret_cast<uint4*>(sb + swz(row, half * 4 + q)) = o.raw;
            }
        }
    };

    if (NF4) __syncthreads();   // lut visible
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* sa = smem + cur * 2 * TILE_BYTES;
        const unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t af[4], bf[4];
            const int slot = ks * 4 + l4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + l15;
                af[i] = as_frag<T>(*reinterpret_cast<const uint4*>(sa + swz(row, slot)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + l15;
                bf[j] = as_frag<T>(*reinterpret_cast<const uint4*>(sb + swz(row, slot)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Mfma<T>::run(bf[j], af[i], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+l15, n = ..+4*l4
    T* Cg = (T*)g.C;
    const bool vec_ok = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 7) == 0);
    const T* bias = (const T*)g.bias;
    auto epi = [&](auto acc_c, auto bias_c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + l15;
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

// ---------------------------------------------------------------------------------------------
// XA[M, R] = X[M,K] @ A[R,K]^T, fp32 out (unscaled, un-rounded; the consumer rounds to T like the
// reference's `torch.matmul(X, A.to(dtype))`). 16 rows per block, 4 waves split K, fragments are
// loaded straight from global (X is streamed once; A stays in L2), fixed-order LDS reduction.
template <typename T, int NT>
__global__ void __launch_bounds__(256) lora_xa_kernel(const T* __restrict__ X, int64_t ldx,
                                                      const T* __restrict__ A, int64_t lda_,
                                                      float* __restrict__ out, int64_t ld_out,
                                                      int M, int K, int R, int out_cols) {
    typedef typename Mfma<T>::frag frag_t;
    __shared__ float red[4][16][NT * 16 + 1];   // NT <= 12 -> <= 49.4 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int m0 = blockIdx.x * 16;
    f32x4_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (K + 31) / 32;
    const int m = m0 + l15;
    for (int s = wave; s < nsteps; s += 4) {
        const int k = s * 32 + l4 * 8;
        uint4 xr = make_uint4(0, 0, 0, 0);
        if (m < M && k < K) xr = *reinterpret_cast<const uint4*>(X + (int64_t)m * ldx + k);
        const frag_t xf = as_frag<T>(xr);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int r = j * 16 + l15;
            uint4 ar = make_uint4(0, 0, 0, 0);
            if (r < R && k < K) ar = *reinterpret_cast<const uint4*>(A + (int64_t)r * lda_ + k);
            // natural order: D[m][r], lane holds r = l15, m = 4*l4 + reg
            acc[j] = Mfma<T>::run(xf, as_frag<T>(ar), acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][l4 * 4 + q][j * 16 + l15] = acc[j][q];
    __syncthreads();
    for (int idx = tid; idx < 16 * out_cols; idx += 256) {
        const int mm = idx / out_cols, c = idx - mm * out_cols;
        if (m0 + mm < M) {
            float v = 0.f;
            if (c < R) v = ((red[0][mm][c] + red[1][mm][c]) + red[2][mm][c]) + red[3][mm][c];
            out[(int64_t)(m0 + mm) * ld_out + c] = v;   // columns R..out_cols-1 are zero padding
        }
    }
}

template <typename T, bool NF4>
int launch_gemm(const GemmArgs& a, int total_tiles, hipStream_t st) {
    const int smem = SMEM_BYTES + (NF4 ? 64 : 0);
    static bool attr_set[64] = {false};   // >48 KiB dynamic LDS needs the opt-in attribute, per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, NF4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, NF4>), dim3((unsigned)total_tiles), dim3(256), smem, st, a);
    return uamd_launch_status();
}

int gemm_entry(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups, int n_groups,
               int accumulate, int nf4, int dtype, void* stream) {
    if (M < 0 || K <= 0 || n_groups < 1 || n_groups > UAMD_GEMM_MAX_GROUPS || !groups) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 7) || (lda & 7) || !aligned16(A)) return UAMD_ERR_ALIGN;
    if (nf4 && (K & 63)) return UAMD_ERR_ARG;
    GemmArgs a;
    a.A = A; a.lda = lda; a.M = M; a.K = K; a.n_groups = n_groups; a.accumulate = accumulate;
    a.tiles_m = (M + BM - 1) / BM;
    int tn = 0;
    for (int i = 0; i < UAMD_GEMM_MAX_GROUPS; ++i) {
        a.tile_start[i] = tn;
        if (i < n_groups) {
            const uamd_gemm_group& g = groups[i];
            if (g.N <= 0 || !g.B || !g.C) return UAMD_ERR_ARG;
            if (!nf4 && ((g.ldb & 7) || !aligned16(g.B))) return UAMD_ERR_ALIGN;
            if (nf4 && (!g.absmax || !aligned16(g.B))) return UAMD_ERR_ARG;
            if (g.lora_xa) {
                if (!g.lora_b || g.R <= 0 || (g.R & 7) || (g.ld_xa & 3) || (g.ld_lb & 7) ||
                    !aligned16(g.lora_xa) || !aligned16(g.lora_b))
                    return UAMD_ERR_ALIGN;
            }
            a.g[i] = g;
            tn += (g.N + BN - 1) / BN;
        } else {
            a.g[i] = groups[0];
        }
    }
    a.tile_start[UAMD_GEMM_MAX_GROUPS] = tn;
    const int64_t total = (int64_t)tn * a.tiles_m;
    if (total > 0x7fffffffLL) return UAMD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) return nf4 ? launch_gemm<bf16_t, true>(a, (int)total, st) : launch_gemm<bf16_t, false>(a, (int)total, st);
    if (dtype == UAMD_F16) return nf4 ? launch_gemm<f16_t, true>(a, (int)total, st) : launch_gemm<f16_t, false>(a, (int)total, st);
    return UAMD_ERR_DTYPE;
}

template <typename T>
int launch_xa(const void* X, int64_t ldx, const void* A, int64_t lda, float* out, int64_t ld_out,
              int M, int K, int R, int out_cols, hipStream_t st) {
    dim3 grid((unsigned)((M + 15) / 16)), block(256);
    const int nt = (R + 15) / 16;
#define L(NT) hipLaunchKernelGGL((lora_xa_kernel<T, NT>), grid, block, 0, st, (const T*)X, ldx, (const T*)A, lda, out, ld_out, M, K, R, out_cols)
    if (nt <= 1) L(1); else if (nt <= 2) L(2); else if (nt <= 3) L(3); else if (nt <= 4) L(4);
    else if (nt <= 6) L(6); else if (nt <= 8) L(8); else if (nt <= 12) L(12);
    else return UAMD_ERR_ARG;
#undef L
    return uamd_launch_status();
}

// Debug probe: records which (row, col) of D each (lane, reg) of v_mfma_f32_16x16x32_bf16 holds.
// out[0][lane][reg] = 1 + row index (from A[i][*] = i+1, B = 1/32), out[1][lane][reg] = 1 + col index.
__global__ void mfma_probe_kernel(float* out) {
    const int lane = threadIdx.x & 63, l15 = lane & 15;
    Vec16<bf16_t> a, b;
    for (int j = 0; j < 8; ++j) { a.e[j] = (bf16_t)(float)(l15 + 1); b.e[j] = (bf16_t)0.03125f; }
    f32x4_t d = Mfma<bf16_t>::run(as_frag<bf16_t>(a.raw), as_frag<bf16_t>(b.raw), f32x4_t{0, 0, 0, 0});
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
    for (int j = 0; j < 8; ++j) { a.e[j] = (bf16_t)0.03125f; b.e[j] = (bf16_t)(float)(l15 + 1); }
    d = Mfma<bf16_t>::run(as_frag<bf16_t>(a.raw), as_frag<bf16_t>(b.raw), f32x4_t{0, 0, 0, 0});
    for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = d[r];
}

}  // namespace

extern "C" int uamd_debug_mfma_probe(float* out, void* stream) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return uamd_launch_status();
}

// C_g[M, N_g] (+)= A[M,K] @ B_g[N_g,K]^T (+ s_g * T(XA_g) @ LB_g^T) for up to 3 groups sharing A.
extern "C" int uamd_gemm_nt(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                            int n_groups, int accumulate, int dtype, void* stream) {
    return gemm_entry(A, lda, M, K, groups, n_groups, accumulate, 0, dtype, stream);
}

// Same contract, B_g given as bitsandbytes NF4 packed bytes ([N_g*K/2]) + fp32 absmax per 64
// elements (blocksize 64); the weight is decoded while it is staged into LDS.
extern "C" int uamd_gemm_nt_nf4(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                                int n_groups, int accumulate, int dtype, void* stream) {
    return gemm_entry(A, lda, M, K, groups, n_groups, accumulate, 1, dtype, stream);
}

// XA[M, out_cols] = X[M,K] @ A[R,K]^T in fp32 (columns R..out_cols-1 written as zeros).
extern "C" int uamd_lora_xa(const void* X, int64_t ldx, const void* A, int64_t lda, float* out,
                            int64_t ld_out, int M, int K, int R, int out_cols, int dtype, void* stream) {
    if (M < 0 || K <= 0 || R <= 0 || out_cols < R || R > 192 || out_cols > 256) return UAMD_ERR_ARG;
    if (M == 0) return UAMD_OK;
    if ((K & 7) || (ldx & 7) || (lda & 7) || !aligned16(X) || !aligned16(A)) return UAMD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16) return launch_xa<bf16_t>(X, ldx, A, lda, out, ld_out, M, K, R, out_cols, st);
    if (dtype == UAMD_F16) return launch_xa<f16_t>(X, ldx, A, lda, out, ld_out, M, K, R, out_cols, st);
    return UAMD_ERR_DTYPE;
}
