// EXPERIMENT (round 2), not built into the library: causal GQA flash-attention forward with 64 q rows per wave, 4 waves per
// block, one wave per SIMD with the 512-register file. It was pasted into csrc/attention.hip (it uses that file's AttnArgs,
// MfmaA, dma16x4g, max_across_halves, pack_pair) and launched with 256 threads / 128 KiB of LDS; the launch stub is at the
// bottom. Result on MI355X (B4 T2048 Hq32 Hk8 D128): 251 us vs 212 us for the shipped 8-wave kernel -- see README.md here.

// ------------------------------------------------------------------------------------------------------------
// Forward with 64 q rows per wave: 4 waves per block, ONE wave per SIMD with the whole 512-register file.
// Why: with 8 waves x 32 rows every wave reads the full K and V tiles from LDS for 32 rows of output -- 256 KB of
// LDS reads per 64-key tile step per CU, half of them 8-byte transposing reads: ~3,000 cycles of LDS pipe per step
// against 2,048 cycles of MFMA (profiles/r01_attn_fwd_trace.txt: tile period 5,800). Here every K / V^T fragment
// read feeds TWO MFMAs (the wave's two 32-row q blocks), which halves the LDS stream. With one wave per SIMD nobody
// else fills the matrix pipe while this wave does its softmax, so the wave pipelines ITSELF over 32-key half tiles
// h = 0, 1, 2, ...:
//     segment h :   MFMA   O^T += V^T(h-1) P^T(h-1)   (16)   then   S^T(h+1) = K(h+1) Q^T   (16)
//                   VALU   softmax of S^T(h) -> P^T(h)          (no dependence on either MFMA group)
// Both streams sit in one basic block; the scores ping-pong between two register sets. The O accumulators (128
// registers) are touched by MFMAs only: the online-softmax rescale is LAZY -- the running max of a row is only
// raised (and O, l rescaled) when the new half tile exceeds it by more than 2^8, decided per wave in a rarely
// taken branch; exp2(s - m_stale) <= 256 keeps fp32 / bf16 range, the final 1/l normalisation is unchanged.
// Same work per block as attn_fwd_kernel (the q tile is 256 / G positions either way), same LDS-DMA tile format
// and swizzles, 4-stage ring (128 KiB): a trip reads V of tile t-1 and t and K of tile t and t+1. Plain causal
// only (no band), G in {1, 2, 4}.
constexpr int NST4 = 4;
constexpr int ATTN_LDS4 = NST4 * STAGE_B;            // 128 KiB
constexpr float LAZY_RESCALE_LOG2 = 8.0f;

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) attn_fwd64_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 64 * (4 / G);
    const int npairs = p.Hk * p.B;
    const int qtile = p.nqt - 1 - (int)(blockIdx.x / npairs);           // heaviest q tiles first
    const int pair_ = (int)(blockIdx.x % npairs);
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int head = kvh * G + (wave % G);
    const int qs = qtile * QT + (wave / G) * 64;                        // first q position of this wave

    // ---- Q^T operand fragments of both 32-row q blocks (lane -> q = qs + 32 qb + l31, 8 d at 16 ks + 8 lh)
    frag_t qf[2][8];
    int q_pos[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q_pos[qb] = qs + 32 * qb + l31;
        const int q_ld = q_pos[qb] < T_ ? q_pos[qb] : T_ - 1;
        const T* qp = (const T*)p.Q + b * p.q_sb + (int64_t)q_ld * p.q_st + (int64_t)head * p.q_sh + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            union { uint4 r; frag_t f; } u;
            u.r = *reinterpret_cast<const uint4*>(qp + ks * 16);
            qf[qb][ks] = u.f;
        }
    }

    // ---- DMA plan: stage = K tile (64 rows x 256 B) then V tile; one DMA instruction = 4 rows; wave w issues
    //      pieces 4w .. 4w+3 of K and of V (swizzles as in attn_fwd_kernel)
    const int nt = min((qtile * QT + QT + KT - 1) / KT, (T_ + KT - 1) / KT);     // tiles the block stages
    int drow[4], dks[4], dvs[4];
    unsigned koff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 4 + (lane >> 4);
        drow[i] = row;
        dks[i] = ((lane & 15) ^ (row & 15)) * 16;
        dvs[i] = ((lane & 15) ^ ((row & 3) << 2)) * 16;
        koff[i] = (unsigned)((int64_t)row * p.k_st * 2 + dks[i]);
        voff[i] = (unsigned)((int64_t)row * p.v_st * 2 + dvs[i]);
    }
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvh * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 4096;
    auto issue = [&](int t, int stage) {                                 // 8 DMA instructions per wave: K's 4, then V's 4
        const int k0 = t * KT;
        const unsigned d = dst_w + stage * STAGE_B;
        if (k0 + KT <= T_) {
            dma16x4g(kbase + (int64_t)k0 * p.k_st, koff[0], koff[1], koff[2], koff[3], d);
            dma16x4g(vbase + (int64_t)k0 * p.v_st, voff[0], voff[1], voff[2], voff[3], d + TILE_B);
        } else {                                                         // ragged last tile: masked rows re-read the last key
            unsigned ko[4], vo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = min(drow[i], T_ - 1 - k0);
                ko[i] = (unsigned)((int64_t)r * p.k_st * 2 + dks[i]);
                vo[i] = (unsigned)((int64_t)r * p.v_st * 2 + dvs[i]);
            }
            dma16x4g(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], ko[2], ko[3], d);
            dma16x4g(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], vo[2], vo[3], d + TILE_B);
        }
    };

    const int kx = l31 & 15;
    const int k_lane = l31 * 256 + (((kx & 14) | (lh ^ (kx & 1))) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int v_lane = (4 * lh + (sg >> 2)) * 256 + ((((sg >> 2) << 2) | (gh << 1) | ((sg >> 1) & 1)) << 4) + (sg & 1) * 8;

    f32x16_t o_acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[i][qb][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    const int nh = 2 * (min(qs + 63, T_ - 1) / KT + 1);                  // half tiles this wave multiplies (<= 2 nt)
    typedef union { uint32_t w[4]; frag_t f; } pfrag_t;                  // P^T operand: 8 keys x this lane's q

    // S^T[32 keys of half h][64 q] = K Q^T: 8 k-steps, every K fragment feeds both q blocks
    auto scores = [&](int h, f32x16_t (&st)[2]) {
        const unsigned char* sk = smem + ((h >> 1) & 3) * STAGE_B + (h & 1) * 32 * 256;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[qb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            union { uint4 r; frag_t f; } u;
            u.r = *reinterpret_cast<const uint4*>(sk + (k_lane ^ (ks * 32)));
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) st[qb] = MfmaA<T>::run(u.f, qf[qb][ks], st[qb]);
        }
    };
    // O^T[d][q] += V^T P^T over the 32 keys of half h: 2 steps of 16 keys, every V^T fragment feeds both q blocks.
    // h = -1 (first segment: nothing to add yet) multiplies the all-zero P of the prologue with tile 0's V.
    auto pv = [&](int h, const pfrag_t (&pf)[2][2]) {
        const int hc = h < 0 ? 0 : h;
        const unsigned char* sv = smem + ((hc >> 1) & 3) * STAGE_B + TILE_B;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int row0 = ((hc & 1) * 2 + c) * 16;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* a0 = sv + row0 * 256 + (v_lane ^ (dt << 6));
                union { s16x4_t hh[2]; frag_t f; } va;
                va.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
                va.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * 256));
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) o_acc[dt][qb] = MfmaA<T>::run(va.f, pf[c][qb].f, o_acc[dt][qb]);
            }
        }
    };
    // softmax, part 1 (before the rare rescale branch): mask if needed, row max of the half tile in the log2 domain
    auto row_max = [&](int h, f32x16_t (&st)[2], float (&mt)[2]) {
        const int k0 = h * 32;
        if ((k0 + 31 > qs) || (k0 + 32 > T_)) {                          // wave-uniform: diagonal / ragged half tile
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > q_pos[qb] || key >= T_) st[qb][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float m = st[qb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, st[qb][r]);
            mt[qb] = max_across_halves(m) * p.scale_log2;
        }
    };
    // the rare branch: raise the running max of both rows, rescale l and the O accumulators
    auto maybe_rescale = [&](const float (&mt)[2]) {
        const bool need = (mt[0] > m_run[0] + LAZY_RESCALE_LOG2) || (mt[1] > m_run[1] + LAZY_RESCALE_LOG2);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(need) != 0, 0)) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float m_new = fmaxf(m_run[qb], mt[qb]);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - (m_new == -INFINITY ? 0.f : m_new));
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i) o_acc[i][qb] *= alpha;
            }
        }
    };
    // softmax, part 2: P = exp2(s c - m) (one fma + one exp per score), row sums, P^T operands in the activation dtype
    auto probs = [&](f32x16_t (&st)[2], pfrag_t (&pf)[2][2]) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float m_ref = m_run[qb] == -INFINITY ? 0.f : m_run[qb];
            float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qb][r], p.scale_log2, -m_ref));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qb][r + 1], p.scale_log2, -m_ref));
                ls0 += e0;
                ls1 += e1;
                pf[r >> 3][qb].w[(r & 7) >> 1] = pack_pair<T>(e0, e1);
            }
            l_run[qb] += ls0 + ls1;
        }
    };

    // ---- prologue: tiles 0, 1 in flight; scores of half tile 0; an all-zero "previous P"
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x16_t sa[2], sb[2];
    pfrag_t pa[2][2], pb[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int j = 0; j < 4; ++j) pb[c][qb].w[j] = 0u;
    scores(0, sa);

    // One segment, the same straight-line code for every half tile: softmax of half h (scores in `cur`) -> `pcur`;
    // meanwhile O += V(h-1) P(h-1) (`pprev`) and the scores of half h+1 -> `nxt`. The very last segment of a wave
    // multiplies a half tile that does not exist (whatever the next ring stage holds; the result is never read):
    // 16 MFMAs per wave, cheaper than a second copy of the loop body.
    auto segment = [&](int h, f32x16_t (&cur)[2], f32x16_t (&nxt)[2], pfrag_t (&pcur)[2][2], const pfrag_t (&pprev)[2][2]) {
        float mt[2];
        row_max(h, cur, mt);
        maybe_rescale(mt);
        pv(h - 1, pprev);
        scores(h + 1, nxt);
        probs(cur, pcur);
    };
    const int nt_w = nh >> 1;
    // tile t+1 (its K half is needed in trip t) has landed for every wave; stage (t+2) & 3 = (t-2) & 3 is free
    auto ring_step = [&](int t) {
        if (t + 1 < nt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 2 < nt) issue(t + 2, (t + 2) & 3);
        }
    };
    for (int t = 0; t < nt_w; ++t) {
        ring_step(t);
        segment(2 * t, sa, sb, pa, pb);
        segment(2 * t + 1, sb, sa, pb, pa);
    }
    pv(nh - 1, pb);
    for (int t = nt_w; t < nt; ++t) ring_step(t);      // G < 4: a wave with an earlier q subtile keeps the ring going

    // ---- epilogue: O = O^T / l, LSE = ln2 * (m + log2 l)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        if (q_pos[qb] < T_) {
            T* op = (T*)p.O + b * p.o_sb + (int64_t)q_pos[qb] * p.o_st + (int64_t)head * p.o_sh;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int d = dt * 32 + qd * 8 + lh * 4;
                    uint2 o;
                    o.x = pack_pair<T>(o_acc[dt][qb][qd * 4 + 0] * inv, o_acc[dt][qb][qd * 4 + 1] * inv);
                    o.y = pack_pair<T>(o_acc[dt][qb][qd * 4 + 2] * inv, o_acc[dt][qb][qd * 4 + 3] * inv);
                    *reinterpret_cast<uint2*>(op + d) = o;
                }
            if (lh == 0)
                p.LSE[((int64_t)b * p.Hq + head) * p.lse_st + q_pos[qb]] = (m_run[qb] + log2f(l_tot)) * 0.6931471805599453f;
        }
    }
}


// ---- launch stub (inside uamd_attn_fwd, before the 8-wave dispatch)
/*
    // 64 q rows per wave (4 waves, one per SIMD): plain causal, G <= 4 (UAMD_TUNE_ATTN_VAR bit 0)
    if (!lo && G <= 4 && (uamd_tuning_get(UAMD_TUNE_ATTN_VAR) & 1)) {
        static bool attr64[2][64] = {{false}};
        if (dtype == UAMD_BF16) {
            if ((rc = set_lds_attr(&attn_fwd64_kernel<bf16_t>, ATTN_LDS4, &attr64[0][dev]))) return rc;
            hipLaunchKernelGGL((attn_fwd64_kernel<bf16_t>), grid, dim3(256), ATTN_LDS4, st, a);
        } else if (dtype == UAMD_F16) {
            if ((rc = set_lds_attr(&attn_fwd64_kernel<f16_t>, ATTN_LDS4, &attr64[1][dev]))) return rc;
            hipLaunchKernelGGL((attn_fwd64_kernel<f16_t>), grid, dim3(256), ATTN_LDS4, st, a);
        } else {
            return UAMD_ERR_DTYPE;
        }
        return uamd_launch_status();
    }
*/
