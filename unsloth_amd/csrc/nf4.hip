// NF4 (bitsandbytes 4-bit NormalFloat, double-quantised absmax) dequantise / quantise.
//
// The reference reaches this arithmetic through ctypes into bitsandbytes' C library
// (third-party, source NOT in /root/reference):
//   unsloth/kernels/utils.py:266-284   symbol binding (cdequantize_blockwise_fp32,
//                                      cdequantize_blockwise_{fp16,bf16}_nf4, ...)
//   unsloth/kernels/utils.py:567-679   fast_dequantize: absmax = code2[absmax_u8]*absmax2 (+offset),
//                                      then W = NF4[nibble] * absmax[j / blocksize]
// bitsandbytes is pinned >=0.45.5 by the reference (pyproject.toml:473); the format restated
// here is its published one (QLoRA paper + bnb functional.dequantize_4bit):
//   * one byte packs two 4-bit codes, HIGH nibble = even element, LOW nibble = odd element,
//     row-major over the flattened [out, in] matrix;
//   * W[j] = LUT[code_j] * absmax_f32[j / blocksize]            (product in fp32, rounded once)
//   * absmax_f32[k] = code2[absmax_u8[k]] * absmax2[k / blocksize2] + offset   (nested quant)
// PARITY UNPINNED against bitsandbytes itself (no bnb in this image, no vectors in the
// reference's tests); pinned against oracle/nf4_ref.c which restates the same format.
//
// The three `cdequantize_blockwise_*` entry points keep bitsandbytes' exact C signatures so the
// reference's ctypes binding (utils.py:272-275) can point at this library unchanged.
//
// HBM-bound: 0.516 B/param read + 2 B/param written. A lane owns 8 elements: one 4-byte load,
// eight LDS LUT reads (16 distinct banks: conflict-free, identical codes broadcast), one fully
// coalesced 16-byte store.
#include "common.h"

namespace {

__constant__ float kNF4[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

// out[i] = code[A[i]] * absmax[i / blocksize]   (8-bit blockwise, bnb kDequantizeBlockwise General8bit)
__global__ void __launch_bounds__(256)
dequant_blockwise8_kernel(const float* __restrict__ code, const uint8_t* __restrict__ A,
                          const float* __restrict__ absmax, float* __restrict__ out, int blocksize,
                          int64_t n, float offset) {
    __shared__ float lut[256];
    lut[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = lut[A[i]] * absmax[i / blocksize] + offset;
}

struct AbsmaxSrc {
    const float* f32;        // direct fp32 absmax (single-quant or pre-dequantised) or NULL
    const uint8_t* u8;       // nested: 8-bit codes
    const float* code2;      // nested: 256-entry map
    const float* absmax2;    // nested: fp32 per blocksize2 codes
    float offset;
    int blocksize2;
};

__device__ __forceinline__ float absmax_at(const AbsmaxSrc& a, const float* code2_lds, int64_t k) {
    if (a.f32) return a.f32[k];
    return code2_lds[a.u8[k]] * a.absmax2[k / a.blocksize2] + a.offset;
}

// 8 elements per lane, row-major output identical in shape to the logical weight.
template <typename T>
__global__ void __launch_bounds__(256)
nf4_dequant_kernel(const uint8_t* __restrict__ packed, AbsmaxSrc am, const float* __restrict__ lut_g,
                   T* __restrict__ out, int64_t n, int blocksize) {
    __shared__ float lut[16];
    __shared__ float code2[256];
    if (threadIdx.x < 16) lut[threadIdx.x] = lut_g ? lut_g[threadIdx.x] : kNF4[threadIdx.x];
    if (am.u8) code2[threadIdx.x] = am.code2[threadIdx.x];
    __syncthreads();
    const int64_t ngroups = n / 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += stride) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(packed + g * 4);
        const int64_t e0 = g * 8;
        float a0 = absmax_at(am, code2, e0 / blocksize);
        float res[8];
        if ((blocksize & 7) == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (w >> (8 * b)) & 0xff;
                res[2 * b] = lut[byte >> 4] * a0;        // high nibble first
                res[2 * b + 1] = lut[byte & 15] * a0;
            }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (w >> (8 * b)) & 0xff;
                res[2 * b] = lut[byte >> 4] * absmax_at(am, code2, (e0 + 2 * b) / blocksize);
                res[2 * b + 1] = lut[byte & 15] * absmax_at(am, code2, (e0 + 2 * b + 1) / blocksize);
            }
        }
        if (sizeof(T) == 2) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.e[j] = from_f32<T>(res[j]);
            st16(out + e0, o);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[e0 + j] = from_f32<T>(res[j]);
        }
    }
    // ragged tail: n not a multiple of 8 (n is always even for packed data, odd handled too)
    if (blockIdx.x == 0) {
        for (int64_t e = ngroups * 8 + threadIdx.x; e < n; e += 256) {
            const uint8_t byte = packed[e >> 1];
            const int c = (e & 1) ? (byte & 15) : (byte >> 4);
            out[e] = from_f32<T>(lut[c] * absmax_at(am, code2, e / blocksize));
        }
    }
}

// The same arithmetic with FOUR groups per lane per trip (blocksize a power of two >= 8, 16-bit output). The one-group
// kernel above keeps a single 4-byte load in flight per lane -- 8 waves x 64 lanes x 4 B x 4 SIMDs x 256 CUs = 2 MB over
// the chip, which at ~2 us of loaded HBM latency is ~1 TB/s of packed codes = ~5 TB/s of total traffic: exactly where it
// measured in the step (60-65 % of HBM, profiles/pmc_traffic.json). Here the four code loads and the four absmax loads of a
// trip are issued before anything waits, every load / store instruction of a wave still covers one contiguous 256 B / 1 KB
// piece, and the block index -> absmax index is a shift (the 64-bit division by a run-time blocksize was ~40 VALU
// instructions per group).
template <typename T>
__global__ void __launch_bounds__(256)
nf4_dequant_x4_kernel(const uint8_t* __restrict__ packed, AbsmaxSrc am, const float* __restrict__ lut_g,
                      T* __restrict__ out, int64_t n, int shift /* log2(blocksize) - 3 */, int blocksize) {
    __shared__ float lut[16];
    __shared__ float code2[256];
    if (threadIdx.x < 16) lut[threadIdx.x] = lut_g ? lut_g[threadIdx.x] : kNF4[threadIdx.x];
    if (am.u8) code2[threadIdx.x] = am.code2[threadIdx.x];
    __syncthreads();
    const int64_t ngroups = n / 8;
    const int64_t ntrips = ngroups / 1024;
    const uint32_t* __restrict__ words = reinterpret_cast<const uint32_t*>(packed);
    auto decode_store = [&](uint32_t w, float a0, int64_t g) {
        Vec16<T> o;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t byte = (w >> (8 * b)) & 0xff;
            o.e[2 * b] = from_f32<T>(lut[byte >> 4] * a0);          // high nibble first
            o.e[2 * b + 1] = from_f32<T>(lut[byte & 15] * a0);
        }
        st16(out + g * 8, o);
    };
    for (int64_t t = blockIdx.x; t < ntrips; t += gridDim.x) {
        const int64_t g0 = t * 1024 + threadIdx.x;
        uint32_t w[4];
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = words[g0 + j * 256];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = absmax_at(am, code2, (g0 + j * 256) >> shift);
#pragma unroll
        for (int j = 0; j < 4; ++j) decode_store(w[j], a[j], g0 + j * 256);
    }
    // the groups past the last whole trip, then the elements past the last whole group (as the one-group kernel)
    for (int64_t g = ntrips * 1024 + (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256)
        decode_store(words[g], absmax_at(am, code2, g >> shift), g);
    if (blockIdx.x == 0) {
        for (int64_t e = ngroups * 8 + threadIdx.x; e < n; e += 256) {
            const uint8_t byte = packed[e >> 1];
            const int c = (e & 1) ? (byte & 15) : (byte >> 4);
            out[e] = from_f32<T>(lut[c] * absmax_at(am, code2, e / blocksize));
        }
    }
}

// Up to FOUR weights in ONE launch (q | k | v, gate | up: the weights of one grouped GEMM, decoded back to back). Alone, a
// [1024, 4096] weight is 512 trips -- a quarter of the chip's resident blocks, 5.5 us at 1.9 TB/s -- and a [4096, 4096] one
// is a single round of one-trip blocks (9.5 us, 4.4 TB/s): latency, not bandwidth. The trips of the segments are dealt
// to the blocks as one sequence (a block's trips are consecutive and may straddle a segment boundary); arithmetic and
// stores are those of nf4_dequant_x4_kernel, so the output is bit-identical to the single launches.
struct DqMulti {
    const uint8_t* packed[4];
    const float* absmax[4];
    const float* lut[4];     // per weight (NULL: the NF4 levels)
    void* out[4];
    int64_t trip_end[4];     // running total of whole trips (numel / 8192) up to and including segment i
    int nseg;
};
template <typename T>
__global__ void __launch_bounds__(256)
nf4_dequant_x4_multi_kernel(DqMulti m, int shift, int64_t per_block) {
    __shared__ float luts[4][16];
    if (threadIdx.x < 64) {
        const float* lg = m.lut[threadIdx.x >> 4];
        luts[threadIdx.x >> 4][threadIdx.x & 15] = lg ? lg[threadIdx.x & 15] : kNF4[threadIdx.x & 15];
    }
    __syncthreads();
    const int64_t total = m.trip_end[m.nseg - 1];
    const int64_t t0 = (int64_t)blockIdx.x * per_block, t1 = min(t0 + per_block, total);
    for (int64_t t = t0; t < t1; ++t) {
        int seg = 0;
        while (t >= m.trip_end[seg]) ++seg;
        const int64_t g0 = (t - (seg ? m.trip_end[seg - 1] : 0)) * 1024 + threadIdx.x;
        const uint32_t* __restrict__ words = reinterpret_cast<const uint32_t*>(m.packed[seg]);
        const float* __restrict__ am = m.absmax[seg];
        const float* lut = luts[seg];
        T* __restrict__ out = (T*)m.out[seg];
        uint32_t w[4];
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = words[g0 + j * 256];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = am[(g0 + j * 256) >> shift];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Vec16<T> o;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (w[j] >> (8 * b)) & 0xff;
                o.e[2 * b] = from_f32<T>(lut[byte >> 4] * a[j]);
                o.e[2 * b + 1] = from_f32<T>(lut[byte & 15] * a[j]);
            }
            st16(out + (g0 + j * 256) * 8, o);
        }
    }
}

// Transposed output: W is logically [rows, cols] (cols contiguous, packed); writes
// out[c * ld_out + r].  Used to hand the backward GEMM (dX = dY @ W) a K-contiguous operand.
// 64x64 tile through LDS; both the packed read and the transposed write are 16B/128B coalesced.
template <typename T>
__global__ void __launch_bounds__(256)
nf4_dequant_t_kernel(const uint8_t* __restrict__ packed, AbsmaxSrc am, const float* __restrict__ lut_g,
                     T* __restrict__ out, int rows, int cols, int64_t ld_out, int blocksize) {
    constexpr int TP = 72;  // padded tile row (elements); 144 B keeps 16-byte alignment
    __shared__ float lut[16];
    __shared__ float code2[256];
    __shared__ __attribute__((aligned(16))) T tile[64 * TP];
    if (threadIdx.x < 16) lut[threadIdx.x] = lut_g ? lut_g[threadIdx.x] : kNF4[threadIdx.x];
    if (am.u8) code2[threadIdx.x] = am.code2[threadIdx.x];
    __syncthreads();
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = threadIdx.x + 256 * it;       // 512 units of 8 elements
        const int r = u >> 3, cq = (u & 7) * 8;
        const int gr = r0 + r, gc = c0 + cq;
        Vec16<T> o;
        if (gr < rows && gc < cols) {
            const int64_t e0 = (int64_t)gr * cols + gc;
            const uint32_t w = *reinterpret_cast<const uint32_t*>(packed + (e0 >> 1));
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (w >> (8 * b)) & 0xff;
                const float a_hi = absmax_at(am, code2, (e0 + 2 * b) / blocksize);
                const float a_lo = absmax_at(am, code2, (e0 + 2 * b + 1) / blocksize);
                o.e[2 * b] = from_f32<T>(lut[byte >> 4] * a_hi);
                o.e[2 * b + 1] = from_f32<T>(lut[byte & 15] * a_lo);
            }
        } else {
            o.raw = make_uint4(0, 0, 0, 0);
        }
        *reinterpret_cast<uint4*>(&tile[r * TP + cq]) = o.raw;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = threadIdx.x + 256 * it;
        const int c = u >> 3, rq = (u & 7) * 8;     // output row c (a column of W), 8 rows of W
        const int gc = c0 + c, gr = r0 + rq;
        if (gc < cols && gr < rows) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.e[j] = tile[(rq + j) * TP + c];
            if (gr + 8 <= rows && ((ld_out & 7) == 0)) {
                st16(out + (int64_t)gc * ld_out + gr, o);
            } else {
                for (int j = 0; j < 8 && gr + j < rows; ++j) out[(int64_t)gc * ld_out + gr + j] = o.e[j];
            }
        }
    }
}

// Transposed output, large-tile version (the one the backward uses): 64 rows x 256 columns per block.
//   * packed read: 32 consecutive lanes cover 128 B of one W row (4 B = 8 codes per lane), two rows per lane;
//   * the lane packs (row r, row r+1) of each of its 8 columns into one 32-bit word and writes it to the
//     transposed LDS tile [256 columns][32 row pairs] with the pair index XOR-ed by (column >> 3): the 32 lanes
//     of a ds_write_b32 hit 32 distinct banks;
//   * transposed write: 8 consecutive lanes cover 128 B of one output row (16 B per lane from one
//     ds_read_b128; the XOR only permutes the four words inside it, undone with register selects).
// 2.5 B/param of HBM traffic, both sides moved in >= 128-byte contiguous segments.
// Measured (profiles/r01_dequant_t_variants.jsonl): this kernel, a version that transposes with
// ds_read_b64_tr_b16, a persistent software-pipelined version and a barrier-free wave-private version (32x128
// tiles, +- non-temporal stores) ALL take the row-major kernel's time plus the same 15-25 us per launch, whatever
// the matrix size (4096x4096 and 14336x4096 alike): the cost is in how the transposed output leaves the chip
// (scattered 128-byte lines written back at the end of the kernel), not in the kernel body. The simplest one stays.
template <typename T>
__global__ void __launch_bounds__(256)
nf4_dequant_t2_kernel(const uint8_t* __restrict__ packed, AbsmaxSrc am, const float* __restrict__ lut_g,
                      T* __restrict__ out, int rows, int cols, int64_t ld_out, int blocksize, int row_fastest) {
    __shared__ float lut[16];
    __shared__ float code2[256];
    __shared__ __attribute__((aligned(16))) uint32_t tile[256 * 32];        // 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) lut[tid] = lut_g ? lut_g[tid] : kNF4[tid];
    if (am.u8) code2[tid] = am.code2[tid];
    __syncthreads();
    // row_fastest: consecutive blocks take consecutive 64-row tiles of W = ADJACENT 128-byte segments of the same 256
    // output rows, so that within a short window whole DRAM pages of the transposed output get written (with the
    // column tile fastest, neighbouring segments of an output row are written by blocks far apart in time)
    const int r0 = (row_fastest ? blockIdx.x : blockIdx.y) * 64, c0 = (row_fastest ? blockIdx.y : blockIdx.x) * 256;
    const int cg = lane & 31;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rp = wave * 2 + (lane >> 5) + 8 * it;         // row pair 0..31
        const int gr = r0 + 2 * rp, gc = c0 + cg * 8;
        uint32_t wA = 0x77777777u, wB = 0x77777777u;            // code 7 = 0.0
        float aA = 0.f, aB = 0.f;
        if (gc < cols) {
            if (gr < rows) {
                const int64_t e0 = (int64_t)gr * cols + gc;
                wA = *reinterpret_cast<const uint32_t*>(packed + (e0 >> 1));
                aA = absmax_at(am, code2, e0 / blocksize);
            }
            if (gr + 1 < rows) {
                const int64_t e1 = (int64_t)(gr + 1) * cols + gc;
                wB = *reinterpret_cast<const uint32_t*>(packed + (e1 >> 1));
                aB = absmax_at(am, code2, e1 / blocksize);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t ba = (wA >> (8 * b)) & 0xff, bb = (wB >> (8 * b)) & 0xff;
            union { T h[2]; uint32_t u; } hi, lo;
            hi.h[0] = from_f32<T>(lut[ba >> 4] * aA);  hi.h[1] = from_f32<T>(lut[bb >> 4] * aB);     // column 2b
            lo.h[0] = from_f32<T>(lut[ba & 15] * aA);  lo.h[1] = from_f32<T>(lut[bb & 15] * aB);     // column 2b+1
            tile[(cg * 8 + 2 * b) * 32 + (rp ^ cg)] = hi.u;
            tile[(cg * 8 + 2 * b + 1) * 32 + (rp ^ cg)] = lo.u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int u = tid + 256 * it;
        const int c = u >> 3, k = u & 7;                        // column c, W rows 8k..8k+7
        const int ccg = c >> 3;
        const uint4 w = *reinterpret_cast<const uint4*>(&tile[c * 32 + ((k ^ (ccg >> 2)) << 2)]);
        const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
        const int x = ccg & 3;
        uint4 o;
        o.x = x == 0 ? wv[0] : x == 1 ? wv[1] : x == 2 ? wv[2] : wv[3];            // o[i] = w[i ^ x]
        o.y = x == 0 ? wv[1] : x == 1 ? wv[0] : x == 2 ? wv[3] : wv[2];
        o.z = x == 0 ? wv[2] : x == 1 ? wv[3] : x == 2 ? wv[0] : wv[1];
        o.w = x == 0 ? wv[3] : x == 1 ? wv[2] : x == 2 ? wv[1] : wv[0];
        const int gc = c0 + c, gr = r0 + 8 * k;
        if (gc < cols && gr < rows) {
            T* dst = out + (int64_t)gc * ld_out + gr;
            if (gr + 8 <= rows && ((ld_out & 7) == 0)) {
                *reinterpret_cast<uint4*>(dst) = o;
            } else {
                union { uint4 q; T e[8]; } v;
                v.q = o;
                for (int j = 0; j < 8 && gr + j < rows; ++j) dst[j] = v.e[j];
            }
        }
    }
}

// Blockwise NF4 quantiser: absmax per block, nearest code by the midpoint decision
// boundaries (strict '>' like bnb's dQuantizeNF4). 8 elements per lane, blocksize/8 lanes
// cooperate through xor-shuffles, so blocksize must be a power of two in [8, 512].
template <typename T>
__global__ void __launch_bounds__(256)
nf4_quant_kernel(const T* __restrict__ in, uint8_t* __restrict__ packed, float* __restrict__ absmax,
                 int64_t n, int blocksize) {
    __shared__ float thr[15];
    if (threadIdx.x < 15) thr[threadIdx.x] = 0.5f * (kNF4[threadIdx.x] + kNF4[threadIdx.x + 1]);
    __syncthreads();
    const int64_t ngroups = (n + 7) / 8;
    const int lanes = blocksize / 8;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v[8];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t e = g * 8 + j;
        v[j] = (g < ngroups && e < n) ? to_f32(in[e]) : 0.f;
        m = fmaxf(m, fabsf(v[j]));
    }
    for (int o = 1; o < lanes; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (g >= ngroups) return;
    if ((threadIdx.x & (lanes - 1)) == 0) absmax[(g * 8) / blocksize] = m;
    const float inv = 1.0f / m;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int code = 7;
        if (m > 0.f) {
            const float x = v[j] * inv;
            code = 0;
#pragma unroll
            for (int t = 0; t < 15; ++t) code += (x > thr[t]) ? 1 : 0;
        }
        const int byte = j >> 1;
        const int shift = 8 * byte + ((j & 1) ? 0 : 4);   // even element -> high nibble
        w |= (uint32_t)code << shift;
    }
    const int64_t e0 = g * 8;
    if (e0 + 8 <= n) {
        *reinterpret_cast<uint32_t*>(packed + g * 4) = w;
    } else {
        for (int b = 0; b < 4 && e0 + 2 * b < n; ++b) packed[g * 4 + b] = (uint8_t)(w >> (8 * b));
    }
}

inline unsigned grid_cap(int64_t work_items) {
    int64_t blocks = (work_items + 255) / 256;
    if (blocks < 1) blocks = 1;
    return (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);
}

template <typename T>
int launch_dequant(const uint8_t* packed, const AbsmaxSrc& am, const float* lut, void* out,
                   int64_t rows, int64_t cols, int blocksize, int transpose, int64_t ld_out,
                   hipStream_t st) {
    const int64_t n = rows * cols;
    if (!transpose) {
        const bool pow2 = blocksize >= 8 && (blocksize & (blocksize - 1)) == 0;
        if (sizeof(T) == 2 && pow2 && n >= 8192 && uamd_tuning_get(UAMD_TUNE_DEQUANT_X4) != 0) {
            // whole trips spread EVENLY over at most 8 blocks per CU (7168 trips of a [14336, 4096] weight = 1792 blocks
            // of 4 trips, not 2048 blocks of which half run a fourth trip alone)
            const int64_t ntrips = n / 8192;
            const int64_t per_block = (ntrips + 2047) / 2048;
            const unsigned grid = (unsigned)((ntrips + per_block - 1) / per_block);
            int shift = 0;
            while ((8 << shift) < blocksize) ++shift;
            hipLaunchKernelGGL((nf4_dequant_x4_kernel<T>), dim3(grid), dim3(256), 0, st, packed, am, lut, (T*)out, n,
                               shift, blocksize);
        } else {
            hipLaunchKernelGGL((nf4_dequant_kernel<T>), dim3(grid_cap(n / 8)), dim3(256), 0, st, packed,
                               am, lut, (T*)out, n, blocksize);
        }
    } else {
        if (sizeof(T) != 2 || (cols & 7)) return UAMD_ERR_ARG;
        const int tv = uamd_tuning_get(UAMD_TUNE_DEQUANT_T);
        if ((blocksize & 7) == 0 && tv != 0) {
            const int rf = tv == 2;
            dim3 grid((unsigned)((cols + 255) / 256), (unsigned)((rows + 63) / 64));
            if (rf) grid = dim3((unsigned)((rows + 63) / 64), (unsigned)((cols + 255) / 256));
            hipLaunchKernelGGL((nf4_dequant_t2_kernel<T>), grid, dim3(256), 0, st, packed, am, lut,
                               (T*)out, (int)rows, (int)cols, ld_out, blocksize, rf);
        } else {
            dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
            hipLaunchKernelGGL((nf4_dequant_t_kernel<T>), grid, dim3(256), 0, st, packed, am, lut,
                               (T*)out, (int)rows, (int)cols, ld_out, blocksize);
        }
    }
    return uamd_launch_status();
}

int dequant_entry(const uint8_t* packed, const AbsmaxSrc& am, const float* lut, void* out,
                  int64_t rows, int64_t cols, int blocksize, int out_dtype, int transpose,
                  int64_t ld_out, void* stream) {
    if (rows < 0 || cols < 0 || blocksize <= 0) return UAMD_ERR_ARG;
    if (rows * cols == 0) return UAMD_OK;
    if (!aligned16(out) || (reinterpret_cast<uintptr_t>(packed) & 3)) return UAMD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (transpose) {
        if (out_dtype == UAMD_BF16) return launch_dequant<bf16_t>(packed, am, lut, out, rows, cols, blocksize, 1, ld_out, st);
        if (out_dtype == UAMD_F16) return launch_dequant<f16_t>(packed, am, lut, out, rows, cols, blocksize, 1, ld_out, st);
        return UAMD_ERR_DTYPE;
    }
    UAMD_DISPATCH_FLOAT(out_dtype, return (launch_dequant<T>(packed, am, lut, out, rows, cols, blocksize, 0, ld_out, st)))
    return UAMD_ERR_DTYPE;
}

}  // namespace

// ---- bitsandbytes-compatible entry points (same C signatures; utils.py:272-275) ----------
extern "C" void cdequantize_blockwise_fp32(float* code, unsigned char* A, float* absmax, float* out,
                                           int blocksize, const int n, void* stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(dequant_blockwise8_kernel, dim3(grid_cap(n)), dim3(256), 0, (hipStream_t)stream,
                       code, A, absmax, out, blocksize, (int64_t)n, 0.0f);
}
static void bnb_nf4(float* code, unsigned char* A, float* absmax, void* out, int blocksize, int n,
                    int dtype, void* stream) {
    AbsmaxSrc am{absmax, nullptr, nullptr, nullptr, 0.f, 1};
    dequant_entry(A, am, code, out, 1, n, blocksize, dtype, 0, 0, stream);
}
extern "C" void cdequantize_blockwise_bf16_nf4(float* code, unsigned char* A, float* absmax, void* out,
                                               int blocksize, const int n, void* stream) {
    bnb_nf4(code, A, absmax, out, blocksize, n, UAMD_BF16, stream);
}
extern "C" void cdequantize_blockwise_fp16_nf4(float* code, unsigned char* A, float* absmax, void* out,
                                               int blocksize, const int n, void* stream) {
    bnb_nf4(code, A, absmax, out, blocksize, n, UAMD_F16, stream);
}
extern "C" void cdequantize_blockwise_fp32_nf4(float* code, unsigned char* A, float* absmax, void* out,
                                               int blocksize, const int n, void* stream) {
    bnb_nf4(code, A, absmax, out, blocksize, n, UAMD_F32, stream);
}

// ---- native entry points -------------------------------------------------------------------
// 8-bit blockwise dequant with the nested-quant offset folded in:
//   out[i] = code[A[i]] * absmax[i / blocksize] + offset
extern "C" int uamd_dequantize_absmax(const float* code2, const uint8_t* absmax_u8, const float* absmax2,
                                      float offset, float* out, int blocksize2, int64_t n, void* stream) {
    if (n < 0 || blocksize2 <= 0) return UAMD_ERR_ARG;
    if (n == 0) return UAMD_OK;
    hipLaunchKernelGGL(dequant_blockwise8_kernel, dim3(grid_cap(n)), dim3(256), 0, (hipStream_t)stream,
                       code2, absmax_u8, absmax2, out, blocksize2, n, offset);
    return uamd_launch_status();
}

// One-launch NF4 dequant of a [rows, cols] weight. Either absmax_f32 (plain / pre-dequantised
// statistics) or the nested triple (absmax_u8, code2, absmax2, offset, blocksize2) is given.
// nf4_lut may be NULL (built-in table). transpose_out=1 writes out[c*ld_out + r] (16-bit dtypes).
extern "C" int uamd_nf4_dequantize(const uint8_t* packed, const float* absmax_f32,
                                   const uint8_t* absmax_u8, const float* code2, const float* absmax2,
                                   float offset, int blocksize2, const float* nf4_lut, void* out,
                                   int64_t rows, int64_t cols, int blocksize, int out_dtype,
                                   int transpose_out, int64_t ld_out, void* stream) {
    if (!absmax_f32 && !(absmax_u8 && code2 && absmax2 && blocksize2 > 0)) return UAMD_ERR_ARG;
    AbsmaxSrc am{absmax_f32, absmax_u8, code2, absmax2, offset, blocksize2 > 0 ? blocksize2 : 1};
    return dequant_entry(packed, am, nf4_lut, out, rows, cols, blocksize, out_dtype, transpose_out,
                         ld_out, stream);
}

// Blockwise NF4 quantiser (first-level only; the 8-bit nested quantisation of absmax is host
// logic, see unsloth_amd/nf4.py). blocksize: power of two in [8, 512].
// Row-major NF4 decode of up to four weights in one launch (see nf4_dequant_x4_multi_kernel). Every weight: fp32 absmax
// (first level, already dequantised), numel a multiple of 8192, 16-bit output, contiguous. Bit-identical to uamd_nf4_dequantize.
extern "C" int uamd_nf4_dequantize_multi(int nseg, const uint8_t* const* packed, const float* const* absmax_f32,
                                         void* const* out, const int64_t* numel, const float* const* nf4_lut, int blocksize,
                                         int out_dtype, void* stream) {
    if (nseg < 1 || nseg > 4 || !packed || !absmax_f32 || !out || !numel) return UAMD_ERR_ARG;
    if (blocksize < 8 || blocksize > 8192 || (blocksize & (blocksize - 1))) return UAMD_ERR_ARG;
    if (out_dtype != UAMD_BF16 && out_dtype != UAMD_F16) return UAMD_ERR_DTYPE;
    DqMulti m;
    int64_t total = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < nseg) {
            if (!packed[i] || !absmax_f32[i] || !out[i] || numel[i] <= 0 || (numel[i] & 8191)) return UAMD_ERR_ARG;
            if (!aligned16(packed[i]) || !aligned16(out[i])) return UAMD_ERR_ALIGN;
            total += numel[i] / 8192;
        }
        m.packed[i] = i < nseg ? packed[i] : nullptr;
        m.absmax[i] = i < nseg ? absmax_f32[i] : nullptr;
        m.lut[i] = (i < nseg && nf4_lut) ? nf4_lut[i] : nullptr;
        m.out[i] = i < nseg ? out[i] : nullptr;
        m.trip_end[i] = total;
    }
    m.nseg = nseg;
    int shift = 0;
    while ((8 << shift) < blocksize) ++shift;
    // whole trips spread evenly over at most 8 blocks per CU, as the single launch does
    const int64_t per_block = (total + 2047) / 2048;
    const unsigned grid = (unsigned)((total + per_block - 1) / per_block);
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == UAMD_BF16)
        hipLaunchKernelGGL((nf4_dequant_x4_multi_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, m, shift, per_block);
    else
        hipLaunchKernelGGL((nf4_dequant_x4_multi_kernel<f16_t>), dim3(grid), dim3(256), 0, st, m, shift, per_block);
    return uamd_launch_status();
}

extern "C" int uamd_nf4_quantize(const void* in, uint8_t* packed, float* absmax, int64_t n,
                                 int blocksize, int in_dtype, void* stream) {
    if (n < 0 || blocksize < 8 || blocksize > 512 || (blocksize & (blocksize - 1))) return UAMD_ERR_ARG;
    if (n % blocksize) return UAMD_ERR_ARG;
    if (n == 0) return UAMD_OK;
    const int64_t ngroups = (n + 7) / 8;
    dim3 grid((unsigned)((ngroups + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    UAMD_DISPATCH_FLOAT(in_dtype, hipLaunchKernelGGL((nf4_quant_kernel<T>), grid, block, 0, st,
                                                     (const T*)in, packed, absmax, n, blocksize))
    return uamd_launch_status();
}
