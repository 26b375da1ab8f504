/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY. Plain-C restatement of the bitsandbytes NF4 byte format
 * (third-party dependency of the reference, pinned bitsandbytes>=0.45.5, pyproject.toml:473; the reference
 * reaches it through ctypes at unsloth/kernels/utils.py:266-284, 650-675).  PARITY UNPINNED against
 * bitsandbytes itself: no bitsandbytes in this image and no dequant vectors in the reference's tests.
 *
 *   W[j]          = NF4[code_j] * absmax_f32[j / blocksize]        high nibble = even element
 *   absmax_f32[k] = code2[absmax_u8[k]] * absmax2[k / blocksize2] + offset
 *
 * Built by __graft_entry__.build() / oracle/Makefile into oracle/_build/libnf4_ref.so; checked against the
 * numpy restatement (oracle/ref_ops.py) in tests/test_oracle_nf4_c.py and used by the GPU parity tests. */
#include <stdint.h>
#include <math.h>

static const float NF4[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

void nf4_ref_dequant_absmax(const uint8_t* absmax_u8, const float* code2, const float* absmax2, float offset,
                            int blocksize2, int64_t n, float* out) {
    for (int64_t k = 0; k < n; ++k) {
        volatile float prod = code2[absmax_u8[k]] * absmax2[k / blocksize2];   /* no fma contraction */
        out[k] = prod + offset;
    }
}

void nf4_ref_dequant(const uint8_t* packed, const float* absmax, const float* lut, int blocksize, int64_t n,
                     float* out) {
    if (!lut) lut = NF4;
    for (int64_t j = 0; j < n; ++j) {
        const uint8_t b = packed[j >> 1];
        const int code = (j & 1) ? (b & 15) : (b >> 4);
        out[j] = lut[code] * absmax[j / blocksize];
    }
}

/* first-level quantiser: absmax per block, x * (1/absmax), nearest code with strict '>' midpoints */
void nf4_ref_quant(const float* in, int blocksize, int64_t n, uint8_t* packed, float* absmax) {
    float thr[15];
    for (int t = 0; t < 15; ++t) thr[t] = 0.5f * (NF4[t] + NF4[t + 1]);
    for (int64_t b0 = 0; b0 < n; b0 += blocksize) {
        float m = 0.f;
        for (int64_t j = b0; j < b0 + blocksize && j < n; ++j) m = fmaxf(m, fabsf(in[j]));
        absmax[b0 / blocksize] = m;
        const float inv = 1.0f / (m > 0.f ? m : 1.0f);
        for (int64_t j = b0; j < b0 + blocksize && j < n; ++j) {
            int code = 7;
            if (m > 0.f) {
                volatile float x = in[j] * inv;
                code = 0;
                for (int t = 0; t < 15; ++t) code += (x > thr[t]);
            }
            if (j & 1) packed[j >> 1] = (uint8_t)((packed[j >> 1] & 0xF0) | code);
            else packed[j >> 1] = (uint8_t)(code << 4);
        }
    }
}
