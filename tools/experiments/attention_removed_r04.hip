// NOT BUILT. Attention kernels that lost their A/B and were taken out of csrc/attention.hip in round 4,
// kept as a record of what was tried; each block is the kernel as it last shipped (helpers it shared with the surviving
// kernels -- AttnArgs, MfmaA, dma16x2, pack_pair2, store_rows_x4 ... -- are in csrc/attention.hip).
//   attn_fwd_pp_kernel   : ping-pong forward, one block per work item (superseded by the persistent form attn_fwd_ps_kernel;
//                          phase trace: profiles/r04_attn_fwd_pp_phase_and_block_timeline.txt)
//   attn_fwd64_kernel    : forward, 4 waves x 64 q rows, hidden AGPR accumulators (parity with the 8-wave kernel, r02)
//   attn_bwd_dkdv_kernel : round-1 dK/dV kernel, 8 waves x 32 keys (superseded by attn_bwd_dkdv4_kernel, r03)
//   attn_bwd_dq4_kernel  : dQ, 4 waves x 64 query rows (parity with the 8-wave kernel, r03)
// ------------------------------------------------------------------------------------------------------------
// Forward, PING-PONG schedule (round 4; the default). Same tiling, LDS image, swizzles and DMA ring as attn_fwd_kernel --
// what changes is WHEN each wave does what. In the lockstep kernel all 8 waves run [K reads + S MFMAs | softmax | V reads +
// PV MFMAs] behind one barrier per tile: the two waves of a SIMD want the matrix pipe in the same phases and leave it idle
// in the same phases, and every MFMA waits for its own LDS read (profiles/r01_attn_fwd_trace.txt: tile period 5,800 cycles
// against 2,048 cycles of MFMA issue). Here a tile is four phases separated by block barriers,
//     P1  K rows of the tile LDS -> 64 registers (16 ds_read_b128), LDS-DMA of the tile two ahead
//     P2  S^T = K Q^T: 16 MFMAs on registers only
//     P3  V^T fragments LDS -> the SAME 64 registers (32 ds_read_b64_tr_b16), online softmax on S^T, P packed
//     P4  O^T += V^T P^T: 16 MFMAs on registers only
// and waves 4-7 (the second wave of every SIMD: a workgroup's waves go to the SIMDs in cyclic order) run ONE PHASE BEHIND
// waves 0-3: a matrix phase (P2, P4) of one wave always sits beside a load / VALU phase (P1, P3) of its SIMD partner -- the
// regime MI355X_MICROARCH.md "Two waves per SIMD" describes (matrix beside memory, never matrix beside matrix). The ring
// stays safe under the skew: the stage of tile t is last read in the trailing group's P3(t), which ends at the barrier
// before the leading group's P1(t + 1) -- the first phase that issues a DMA (tile t + 3) into that stage; every wave waits
// for ITS pieces of tile t + 1 at the end of its P3(t), one barrier before anybody reads them.
template <typename T, bool BAND>
__global__ void __launch_bounds__(512, 2) attn_fwd_pp_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
#ifdef UAMD_ATTN_TRACE
    const unsigned long long tb_start = __builtin_amdgcn_s_memtime();
    unsigned long long tb_loop0 = 0, tb_loop1 = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int QT = 32 * p.nsub;
    const int npairs = p.Hk * p.B;
    int rank_, pair_;
    block_to_work((int)blockIdx.x, p.nqt, npairs, p.xcd_map, rank_, pair_);
    const int qtile = p.nqt - 1 - rank_;                              // heaviest q tiles first
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int head = kvh * G + (wave % G);
    const int qs = qtile * QT + (wave / G) * 32;
    const int q_pos = qs + l31;
    const int q_ld = q_pos < T_ ? q_pos : T_ - 1;
    const int lo_q = BAND ? p.lo[(int64_t)b * T_ + q_ld] : 0;
    const int lo_w0 = BAND ? __builtin_amdgcn_readfirstlane(lo_q) : 0, lo_w1 = BAND ? __builtin_amdgcn_readlane(lo_q, 31) : 0;
    const int t_first = BAND ? p.lo[(int64_t)b * T_ + min(qtile * QT, T_ - 1)] / KT : 0;

    const int nkv_blk = min((qtile * QT + QT + KT - 1) / KT, (T_ + KT - 1) / KT);
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvh * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 2048;
    // DMA plan: a stage = K tile (64 rows x 256 B) then V tile; one instruction = 4 rows; wave w issues pieces 2w, 2w + 1
    // (rows 8w .. 8w + 7) of K and of V; lane -> (row = 4 piece + (lane >> 4), stored slot = lane & 15), the stored slot
    // holds logical slot s ^ (row & 15) (K) / s ^ ((row & 3) << 2) (V). The four per-lane source offsets are RECOMPUTED at
    // every issue from an opaque copy of the lane id (a dozen VALU instructions per tile): kept in registers across the tile
    // loop they are the first thing hipcc spills, and a scratch reload in front of the DMA drains vmcnt (= the ring)
    auto issue = [&](int t, int stage) {
#if defined(UAMD_ATTN_DBG) && UAMD_ATTN_DBG == 1
        if (t >= 0) return;                        // timing experiment: no K / V traffic at all (results are garbage)
        const int k0 = 0;
#elif defined(UAMD_ATTN_DBG) && UAMD_ATTN_DBG == 2
        const int k0 = 0;                          // timing experiment: every tile is tile 0 (L2-resident)
#else
        const int k0 = t * KT;
#endif
        const unsigned d = dst_w + stage * STAGE_B;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int rmax = T_ - 1 - k0;                                  // ragged last tile: rows past the end re-read the last key
        unsigned ko[2], vo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 4 + (ln >> 4);
            const int r = min(row, rmax);
            ko[i] = (unsigned)(r * (int)p.k_st * 2 + ((ln & 15) ^ (row & 15)) * 16);
            vo[i] = (unsigned)(r * (int)p.v_st * 2 + ((ln & 15) ^ ((row & 3) << 2)) * 16);
        }
        dma16x2(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], d, d + 1024);
        dma16x2(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], d + TILE_B, d + TILE_B + 1024);
    };
    const int kx = l31 & 15;
    const int k_lane = l31 * 256 + (((kx & 14) | (lh ^ (kx & 1))) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int v_lane = (4 * lh + (sg >> 2)) * 256 + ((((sg >> 2) << 2) | (gh << 1) | ((sg >> 1) & 1)) << 4) + (sg & 1) * 8;

    f32x16_t o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = nkv_blk - t_first;
    const int last_tile_wave = min(qs + 31, T_ - 1) / KT;
    const int first_tile_wave = lo_w0 / KT;
    const int t_pre_end = min(nkv_blk, (lo_w1 + KT - 1) / KT);       // tiles that start below the band edge
    const int t_diag = (qs + 1) / KT, t_rag = (T_ % KT) ? T_ / KT : nkv_blk;
    const int t_suf = max(t_pre_end, min(min(t_diag, t_rag), nkv_blk));

    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed for everybody; the trailing group then drops one phase behind
    issue(t_first, 0);
    if (nt > 1) issue(t_first + 1, 1);
    // Q^T operand fragments (B operand: lane -> q = l31, 8 d at 16 ks + 8 lh), resident for the whole tile loop. Loaded
    // AFTER the first tiles' DMA was issued and waited for with a wait the COMPILER can see (the builtin, not asm): hipcc
    // counts only its own loads, and would otherwise thread a vmcnt(11) ... vmcnt(4) countdown through the first S MFMAs of
    // every tile -- which on the hardware's single counter also waits for LDS-DMA pieces that are meant to stay in flight
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
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): Q fragments + tiles 0 and 1 (tile 1 is needed four phases on)
    bar();
    if (wave >= 4) bar();
#ifdef UAMD_ATTN_TRACE
    tb_loop0 = __builtin_amdgcn_s_memtime();
#endif

#ifdef UAMD_ATTN_TRACE
    // deferred stamps (tile 8 only): s_memtime lands in SGPRs and is consumed after the phase's own waits, so a stamp
    // between "reads issued" and "reads waited for" does not itself wait for the reads
    unsigned long long tsx[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tsx[i] = 0;
#define PSTAMP(I) do { __builtin_amdgcn_sched_barrier(0); if (ti == 8) tsx[I] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(I) do { } while (0)
#endif
    frag_t kv[16];                       // P1 -> P2: K rows (kt * 8 + ks);  P3 -> P4: V^T fragments (u * 4 + dt)
    f32x16_t st[2];
    frag_t pb[4];
    // P^T fragment of 16-key step u = 2 kt + c (registers 8 c .. 8 c + 7 of st[kt]): exp2, partial row sums, packing
    auto softmax_piece = [&](int u, float m_ref, float& ls0, float& ls1) {
        const int kt = u >> 1, c = u & 1;
        union { uint32_t w[4]; frag_t f; } w_;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][8 * c + 2 * jj], p.scale_log2, -m_ref));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][8 * c + 2 * jj + 1], p.scale_log2, -m_ref));
            ls0 += e0;
            ls1 += e1;
            w_.w[jj] = pack_pair2<T>(e0, e1);
        }
        pb[u] = w_.f;
    };
    for (int ti = 0; ti < nt; ++ti) {
        const int t = t_first + ti;
        const bool live = !(t > last_tile_wave || t < first_tile_wave);         // wave-uniform
        const unsigned char* sk = smem + (ti % NST) * STAGE_B;
        const unsigned char* sv = sk + TILE_B;
        // ---------------- P1 (load): K rows -> 64 registers
        PSTAMP(0);
        if (live) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    union { uint4 r; frag_t f; } u;
                    u.r = *reinterpret_cast<const uint4*>(sk + kt * 32 * 256 + (k_lane ^ (ks * 32)));
                    kv[kt * 8 + ks] = u.f;
                }
            PSTAMP(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        PSTAMP(2);
        bar();
        PSTAMP(3);
        // ---------------- P2 (matrix): S^T[key][q] = K Q^T on registers; the LDS-DMA of tile t + 2 rides in the MFMA shadow
        //                  (its stage was last read two barriers ago; an issue costs ~60 cycles among bare MFMAs, 100-185 in a
        //                  phase that also carries LDS reads -- MI355X_MICROARCH.md)
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[0][r] = 0.f; st[1][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st[0] = MfmaA<T>::run(kv[ks], qf[ks], st[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ti + 2 < nt) issue(t + 2, (ti + 2) % NST);
        if (live) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 4; ks < 8; ++ks) st[0] = MfmaA<T>::run(kv[ks], qf[ks], st[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) st[1] = MfmaA<T>::run(kv[8 + ks], qf[ks], st[1]);
        }
        PSTAMP(4);
        bar();
        PSTAMP(5);
        // ---------------- P3 (load + row statistics): V^T fragments -> the same 64 registers; mask, row max (a TREE: a serial
        //                  fmax chain is 32 dependent VALU latencies), rescale test; P^T of the first 16-key step
        float m_ref = 0.f, ls0 = 0.f, ls1 = 0.f;
        if (live) {
            const int k0 = t * KT;
            const bool need_mask = BAND ? (t < t_pre_end || t >= t_suf) : (t >= t_suf);
            if (need_mask) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key > q_pos || key >= T_ || key < lo_q) st[kt][r] = -INFINITY;
                    }
            }
            float m8[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int a0 = (u * 16) * 256 + (v_lane ^ (dt << 6));
                    union { s16x4_t h[2]; frag_t f; } va;
                    va.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0));
                    va.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sv + a0 + 8 * 256));
                    kv[u * 4 + dt] = va.f;
                }
                // leaves of the max tree for 8 scores, issued between the read bursts
                const int kt = u >> 1, c = u & 1;
                m8[2 * u] = fmaxf(fmaxf(st[kt][8 * c], st[kt][8 * c + 1]), fmaxf(st[kt][8 * c + 2], st[kt][8 * c + 3]));
                m8[2 * u + 1] = fmaxf(fmaxf(st[kt][8 * c + 4], st[kt][8 * c + 5]), fmaxf(st[kt][8 * c + 6], st[kt][8 * c + 7]));
            }
            PSTAMP(6);
            float mt = fmaxf(fmaxf(fmaxf(m8[0], m8[1]), fmaxf(m8[2], m8[3])), fmaxf(fmaxf(m8[4], m8[5]), fmaxf(m8[6], m8[7])));
            mt = max_across_halves(mt) * p.scale_log2;
            if (__builtin_amdgcn_ballot_w64(mt > m_run) != 0) {
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __builtin_amdgcn_exp2f(m_run - (m_new == -INFINITY ? 0.f : m_new));
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i) o_acc[i] *= alpha;
            }
            m_ref = m_run == -INFINITY ? 0.f : m_run;
            softmax_piece(0, m_ref, ls0, ls1);
            __builtin_amdgcn_sched_barrier(0);       // (nothing above may sink below the wait: it is what hides the reads)
            PSTAMP(7);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        PSTAMP(8);
        // this wave's pieces of tile t + 1 have landed before anybody reads them (one barrier from now)
        if (ti + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PSTAMP(9);
        bar();
        PSTAMP(10);
        // ---------------- P4 (matrix): O^T[d][q] += V^T P^T on registers; the exponentials of step u + 1 ride in the shadow
        //                  of step u's four MFMAs (one wave hides <= 5 single-issue instructions per 32x32x16 MFMA)
        if (live) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o_acc[dt] = MfmaA<T>::run(kv[u * 4 + dt], pb[u], o_acc[dt]);
                if (u < 3) softmax_piece(u + 1, m_ref, ls0, ls1);
#pragma unroll
                for (int g_ = 0; g_ < 4; ++g_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);       // 7 VALU (28 per step: 8 fma, 8 exp2, 8 add, 4 cvt)
                }
            }
            l_run += ls0 + ls1;
        }
        PSTAMP(11);
        bar();
#ifdef UAMD_ATTN_TRACE
        if (ti == 8) tsx[12] = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef UAMD_ATTN_TRACE
    if (g_attn_trace && lane == 0 && blockIdx.x < 256) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g_attn_trace[(blockIdx.x * 8 + wave) * 16 + i] = (unsigned)tsx[i];
    }
#endif
    if (wave < 4) bar();                 // the leading group's extra barrier = the trailing group's last phase
#ifdef UAMD_ATTN_TRACE
    tb_loop1 = __builtin_amdgcn_s_memtime();
#endif

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    {
        // the row address is formed HERE (an opaque copy of the row index): hoisted to the kernel entry, the 64-bit pointer
        // pair is spilled around the tile loop and its reload's vmcnt(0) drains the LDS-DMA ring
        int qr = q_ld;
        asm volatile("" : "+v"(qr));
        T* op = (T*)p.O + b * p.o_sb + (int64_t)qr * p.o_st + (int64_t)head * p.o_sh;
        store_rows_x4<T>(op, o_acc, inv, lh, q_pos < T_);
        if (lh == 0 && q_pos < T_) p.LSE[((int64_t)b * p.Hq + head) * p.lse_st + qr] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
#ifdef UAMD_ATTN_TRACE
    // block timeline (second region of the trace buffer): entry | tile loop start | tile loop end | stores issued | stores
    // done | HW_ID | XCC_ID, per wave -- tools/attn_trace.py rebuilds every CU's sequence of blocks from it
    if (g_attn_trace) {
        const unsigned long long tb_issued = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tb_done = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x < 4096) {
            unsigned* o_ = g_attn_trace + 32768 + (blockIdx.x * 8 + wave) * 8;
            o_[0] = (unsigned)tb_start; o_[1] = (unsigned)tb_loop0; o_[2] = (unsigned)tb_loop1; o_[3] = (unsigned)tb_issued;
            o_[4] = (unsigned)tb_done;
            o_[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID
            o_[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));       // HW_REG_XCC_ID
            o_[7] = (unsigned)nt;
        }
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------
// Forward with 64 q rows per wave: 4 waves per block, ONE wave per SIMD (512 registers).
// Why: with 8 waves x 32 rows every wave reads the full K and V tiles from LDS for 32 rows of output -- 256 KB of LDS
// reads per 64-key step per CU, half of them 8-byte transposing reads: ~3,000 cycles of LDS pipe against 2,048 cycles
// of MFMA (profiles/r01_attn_fwd_trace.txt: tile period 5,800). Here every K / V^T fragment feeds TWO MFMAs (the
// wave's two 32-row q blocks). With one wave per SIMD nobody else fills the matrix pipe during the softmax, so the wave
// pipelines ITSELF over 32-key half tiles h = 0, 1, 2, ...; segment h is
//     MFMA:  S^T(h+1) = K(h+1) Q^T  (16)   and   O^T += V^T(h-1) P^T(h-1)  (16)
//     VALU:  row max of S^T(h), then  P^T(h) = exp2(S^T(h) c - m)
// cut into chunks of {2-3 MFMAs, the VALU work of 4 scores} fenced with sched_barrier so the order in the source IS
// the issue order. What the compiler must not decide (tools/experiments/README.md: the first version of this kernel
// lost to its register allocation):
//   * the O accumulators live in AGPRs a[0:127] for the whole kernel: their MFMAs are inline asm on PINNED tuples
//     ("+{a[0:15]}" ...), and the only other thing that ever touches them, the online-softmax rescale, is inline asm on
//     the same physical registers, in the slow path only;
//   * the score accumulators are VGPRs (asm MFMAs in the VGPR-destination form): the softmax reads them directly.
// The rescale is LAZY: a row's reference max is raised (and O, l rescaled) only when a half tile exceeds it by more
// than 2^8 -- exp2(s - m_stale) <= 256 keeps fp32 / bf16 range, the 1/l normalisation is unchanged. The fast loop
// contains no rescale code; a wave that needs one leaves the loop after the row max, runs the general segment (which
// also handles the causal / ragged mask of the wave's last tile) and re-enters.
// Same work per block as attn_fwd_kernel (q tile = 256 / G positions), same LDS-DMA tile format and swizzles, 4-stage
// ring (128 KiB): a trip reads V of tiles t-1, t and K of tiles t, t+1; the refill happens mid-trip so that a tile has
// two trips to land. Plain causal (no band), G in {1, 2, 4}.
constexpr int NST4 = 4;
constexpr int ATTN_LDS4 = NST4 * STAGE_B;            // 128 KiB
constexpr float LAZY_RESCALE_LOG2 = 8.0f;

// ---- O accumulators: AGPRs a0..a127, HIDDEN from the compiler (tuple i = a[16 i : 16 i + 15] = accumulator [dt * 2 + qb]).
// They are not C++ values: every instruction that touches them is inline asm naming the physical registers, and every
// such asm clobbers all 128, so the compiler keeps nothing of its own in them across these statements (a value pinned
// only by a "+{a[..]}" constraint still gets copied to VGPRs at every control-flow merge: 128 moves per half tile).
#ifndef UAMD_A64_DBG
#define UAMD_A64_DBG 0
#endif
#if UAMD_A64_DBG & 2
#define UAMD_A64_POST "\n\ts_nop 7\n\ts_nop 7"
#else
#define UAMD_A64_POST ""
#endif
#define UAMD_O_CLOBBER "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
__device__ __forceinline__ void o_zero() {
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" ::: UAMD_O_CLOBBER);
}
template <typename T, int I>
__device__ __forceinline__ void pv_mfma_pinned(typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
    if constexpr (I == 0) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 1) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[16:31], %0, %1, a[16:31]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 2) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[32:47], %0, %1, a[32:47]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 3) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[48:63], %0, %1, a[48:63]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 4) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[64:79], %0, %1, a[64:79]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 5) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[80:95], %0, %1, a[80:95]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 6) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[96:111], %0, %1, a[96:111]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
    else if constexpr (I == 7) {
        if constexpr (std::is_same<T, bf16_t>::value)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
        else
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[112:127], %0, %1, a[112:127]" UAMD_A64_POST :: "v"(a), "v"(b) : UAMD_O_CLOBBER);
    }
}
// tuple I *= alpha through a scratch VGPR; the s_nop runs cover MFMA-write -> accvgpr_read and accvgpr_write -> MFMA-read
template <int I>
__device__ __forceinline__ void scale_pinned(float alpha) {
    float tmp;
    if constexpr (I == 0)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_read_b32 %0, a2\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_read_b32 %0, a3\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_read_b32 %0, a4\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_read_b32 %0, a5\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_read_b32 %0, a6\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_read_b32 %0, a7\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_read_b32 %0, a8\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_read_b32 %0, a9\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_read_b32 %0, a10\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_read_b32 %0, a11\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_read_b32 %0, a12\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_read_b32 %0, a13\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_read_b32 %0, a14\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_read_b32 %0, a15\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a15, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 1)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a16\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_read_b32 %0, a17\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_read_b32 %0, a18\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_read_b32 %0, a19\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a19, %0\n\tv_accvgpr_read_b32 %0, a20\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_read_b32 %0, a21\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_read_b32 %0, a22\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_read_b32 %0, a23\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a23, %0\n\tv_accvgpr_read_b32 %0, a24\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a24, %0\n\tv_accvgpr_read_b32 %0, a25\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a25, %0\n\tv_accvgpr_read_b32 %0, a26\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_read_b32 %0, a27\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a27, %0\n\tv_accvgpr_read_b32 %0, a28\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a28, %0\n\tv_accvgpr_read_b32 %0, a29\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a29, %0\n\tv_accvgpr_read_b32 %0, a30\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_read_b32 %0, a31\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a31, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 2)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a32\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a32, %0\n\tv_accvgpr_read_b32 %0, a33\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_read_b32 %0, a34\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_read_b32 %0, a35\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a35, %0\n\tv_accvgpr_read_b32 %0, a36\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a36, %0\n\tv_accvgpr_read_b32 %0, a37\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_read_b32 %0, a38\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_read_b32 %0, a39\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a39, %0\n\tv_accvgpr_read_b32 %0, a40\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a40, %0\n\tv_accvgpr_read_b32 %0, a41\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a41, %0\n\tv_accvgpr_read_b32 %0, a42\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_read_b32 %0, a43\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a43, %0\n\tv_accvgpr_read_b32 %0, a44\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_read_b32 %0, a45\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a45, %0\n\tv_accvgpr_read_b32 %0, a46\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_read_b32 %0, a47\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a47, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 3)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a48\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a48, %0\n\tv_accvgpr_read_b32 %0, a49\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a49, %0\n\tv_accvgpr_read_b32 %0, a50\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_read_b32 %0, a51\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a51, %0\n\tv_accvgpr_read_b32 %0, a52\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a52, %0\n\tv_accvgpr_read_b32 %0, a53\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a53, %0\n\tv_accvgpr_read_b32 %0, a54\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_read_b32 %0, a55\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a55, %0\n\tv_accvgpr_read_b32 %0, a56\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a56, %0\n\tv_accvgpr_read_b32 %0, a57\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a57, %0\n\tv_accvgpr_read_b32 %0, a58\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_read_b32 %0, a59\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a59, %0\n\tv_accvgpr_read_b32 %0, a60\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a60, %0\n\tv_accvgpr_read_b32 %0, a61\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a61, %0\n\tv_accvgpr_read_b32 %0, a62\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_read_b32 %0, a63\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a63, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 4)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a64\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a64, %0\n\tv_accvgpr_read_b32 %0, a65\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a65, %0\n\tv_accvgpr_read_b32 %0, a66\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_read_b32 %0, a67\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a67, %0\n\tv_accvgpr_read_b32 %0, a68\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a68, %0\n\tv_accvgpr_read_b32 %0, a69\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a69, %0\n\tv_accvgpr_read_b32 %0, a70\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_read_b32 %0, a71\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a71, %0\n\tv_accvgpr_read_b32 %0, a72\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a72, %0\n\tv_accvgpr_read_b32 %0, a73\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a73, %0\n\tv_accvgpr_read_b32 %0, a74\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_read_b32 %0, a75\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a75, %0\n\tv_accvgpr_read_b32 %0, a76\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a76, %0\n\tv_accvgpr_read_b32 %0, a77\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_read_b32 %0, a78\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_read_b32 %0, a79\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a79, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 5)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a80\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a80, %0\n\tv_accvgpr_read_b32 %0, a81\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a81, %0\n\tv_accvgpr_read_b32 %0, a82\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_read_b32 %0, a83\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a83, %0\n\tv_accvgpr_read_b32 %0, a84\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a84, %0\n\tv_accvgpr_read_b32 %0, a85\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a85, %0\n\tv_accvgpr_read_b32 %0, a86\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_read_b32 %0, a87\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a87, %0\n\tv_accvgpr_read_b32 %0, a88\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_read_b32 %0, a89\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a89, %0\n\tv_accvgpr_read_b32 %0, a90\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_read_b32 %0, a91\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a91, %0\n\tv_accvgpr_read_b32 %0, a92\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a92, %0\n\tv_accvgpr_read_b32 %0, a93\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a93, %0\n\tv_accvgpr_read_b32 %0, a94\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_read_b32 %0, a95\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a95, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 6)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a96\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a96, %0\n\tv_accvgpr_read_b32 %0, a97\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a97, %0\n\tv_accvgpr_read_b32 %0, a98\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a98, %0\n\tv_accvgpr_read_b32 %0, a99\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a99, %0\n\tv_accvgpr_read_b32 %0, a100\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a100, %0\n\tv_accvgpr_read_b32 %0, a101\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a101, %0\n\tv_accvgpr_read_b32 %0, a102\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a102, %0\n\tv_accvgpr_read_b32 %0, a103\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a103, %0\n\tv_accvgpr_read_b32 %0, a104\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a104, %0\n\tv_accvgpr_read_b32 %0, a105\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a105, %0\n\tv_accvgpr_read_b32 %0, a106\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a106, %0\n\tv_accvgpr_read_b32 %0, a107\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a107, %0\n\tv_accvgpr_read_b32 %0, a108\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a108, %0\n\tv_accvgpr_read_b32 %0, a109\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a109, %0\n\tv_accvgpr_read_b32 %0, a110\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a110, %0\n\tv_accvgpr_read_b32 %0, a111\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a111, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
    else if constexpr (I == 7)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a112\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a112, %0\n\tv_accvgpr_read_b32 %0, a113\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a113, %0\n\tv_accvgpr_read_b32 %0, a114\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a114, %0\n\tv_accvgpr_read_b32 %0, a115\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a115, %0\n\tv_accvgpr_read_b32 %0, a116\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a116, %0\n\tv_accvgpr_read_b32 %0, a117\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a117, %0\n\tv_accvgpr_read_b32 %0, a118\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a118, %0\n\tv_accvgpr_read_b32 %0, a119\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a119, %0\n\tv_accvgpr_read_b32 %0, a120\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a120, %0\n\tv_accvgpr_read_b32 %0, a121\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a121, %0\n\tv_accvgpr_read_b32 %0, a122\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a122, %0\n\tv_accvgpr_read_b32 %0, a123\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a123, %0\n\tv_accvgpr_read_b32 %0, a124\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a124, %0\n\tv_accvgpr_read_b32 %0, a125\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a125, %0\n\tv_accvgpr_read_b32 %0, a126\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a126, %0\n\tv_accvgpr_read_b32 %0, a127\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a127, %0\n\ts_nop 4" : "=&v"(tmp) : "v"(alpha) : UAMD_O_CLOBBER);
}
// tuple I -> 16 floats (after the last MFMA: the leading s_nops cover its write)
template <int I>
__device__ __forceinline__ void read_pinned(float (&f)[16]) {
    if constexpr (I == 0)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\tv_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 1)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\tv_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\tv_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\tv_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 2)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\tv_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\tv_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\tv_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\tv_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\tv_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 3)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\tv_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\tv_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\tv_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\tv_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\tv_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 4)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\tv_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\tv_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\tv_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\tv_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\tv_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 5)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\tv_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\tv_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\tv_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\tv_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\tv_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 6)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\tv_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99\n\tv_accvgpr_read_b32 %4, a100\n\tv_accvgpr_read_b32 %5, a101\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a103\n\tv_accvgpr_read_b32 %8, a104\n\tv_accvgpr_read_b32 %9, a105\n\tv_accvgpr_read_b32 %10, a106\n\tv_accvgpr_read_b32 %11, a107\n\tv_accvgpr_read_b32 %12, a108\n\tv_accvgpr_read_b32 %13, a109\n\tv_accvgpr_read_b32 %14, a110\n\tv_accvgpr_read_b32 %15, a111" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
    else if constexpr (I == 7)
        asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\tv_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115\n\tv_accvgpr_read_b32 %4, a116\n\tv_accvgpr_read_b32 %5, a117\n\tv_accvgpr_read_b32 %6, a118\n\tv_accvgpr_read_b32 %7, a119\n\tv_accvgpr_read_b32 %8, a120\n\tv_accvgpr_read_b32 %9, a121\n\tv_accvgpr_read_b32 %10, a122\n\tv_accvgpr_read_b32 %11, a123\n\tv_accvgpr_read_b32 %12, a124\n\tv_accvgpr_read_b32 %13, a125\n\tv_accvgpr_read_b32 %14, a126\n\tv_accvgpr_read_b32 %15, a127" : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]), "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]));
}
// score MFMAs with the accumulator in VGPRs (the softmax reads it) and the Q^T operand in AGPRs ("a": the 64 registers
// of Q fragments are MFMA-only, so they stay out of the VGPR file); first k-step with C = 0
template <typename T>
__device__ __forceinline__ void s_mfma_first(f32x16_t& s, typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
#if UAMD_A64_DBG & 1
    f32x16_t z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    s = MfmaA<T>::run(a, b, z);
    return;
#endif
    if constexpr (std::is_same<T, bf16_t>::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" UAMD_A64_POST : "=&v"(s) : "v"(a), "a"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" UAMD_A64_POST : "=&v"(s) : "v"(a), "a"(b));
}
template <typename T>
__device__ __forceinline__ void s_mfma(f32x16_t& s, typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
#if UAMD_A64_DBG & 1
    s = MfmaA<T>::run(a, b, s);
    return;
#endif
#if UAMD_A64_DBG & 32
    return;
#endif
    if constexpr (std::is_same<T, bf16_t>::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" UAMD_A64_POST : "+v"(s) : "v"(a), "a"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" UAMD_A64_POST : "+v"(s) : "v"(a), "a"(b));
}
// the LAST MFMA of a score chain carries the wait states its VALU readers need (XDL write -> VALU read: up to 19): the
// compiler does not know these asm statements are MFMAs and may schedule a read of `s` right behind them
template <typename T>
__device__ __forceinline__ void s_mfma_last(f32x16_t& s, typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
#if UAMD_A64_DBG & 1
    s = MfmaA<T>::run(a, b, s);
    return;
#endif
    if constexpr (std::is_same<T, bf16_t>::value)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(s) : "v"(a), "a"(b));
    else
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(s) : "v"(a), "a"(b));
}
template <typename T>
__device__ __forceinline__ void pv_i(int i, typename MfmaA<T>::frag a, typename MfmaA<T>::frag b) {
#if UAMD_A64_DBG & 8
    return;
#endif
    switch (i) {                                  // i is a constant after unrolling: one case survives
        case 0: pv_mfma_pinned<T, 0>(a, b); break;
        case 1: pv_mfma_pinned<T, 1>(a, b); break;
        case 2: pv_mfma_pinned<T, 2>(a, b); break;
        case 3: pv_mfma_pinned<T, 3>(a, b); break;
        case 4: pv_mfma_pinned<T, 4>(a, b); break;
        case 5: pv_mfma_pinned<T, 5>(a, b); break;
        case 6: pv_mfma_pinned<T, 6>(a, b); break;
        default: pv_mfma_pinned<T, 7>(a, b); break;
    }
}

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
    int rank_, pair_;
    block_to_work((int)blockIdx.x, p.nqt, npairs, p.xcd_map, rank_, pair_);
    const int qtile = p.nqt - 1 - rank_;                              // heaviest q tiles first
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
    const T* kbase = (const T*)p.K + b * p.k_sb + (int64_t)kvh * p.k_sh;
    const T* vbase = (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    const unsigned dst_w = lds_base + wave * 4096;
    auto issue = [&](int t, int stage) {                                 // 8 DMA instructions per wave: K's 4, then V's 4
        const int k0 = t * KT;
        const unsigned d = dst_w + stage * STAGE_B;
        // per-lane source offsets rebuilt per call from an opaque copy of the lane id (a dozen VALU instructions per
        // tile): holding them would cost 8-20 registers for the whole kernel. Ragged last tile: masked rows re-read
        // the last key.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        unsigned ko[4], vo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 4 + (ln >> 4);
            const int r = min(row, T_ - 1 - k0);
            ko[i] = (unsigned)((int64_t)r * p.k_st * 2 + ((ln & 15) ^ (row & 15)) * 16);
            vo[i] = (unsigned)((int64_t)r * p.v_st * 2 + ((ln & 15) ^ ((row & 3) << 2)) * 16);
        }
        dma16x4g(kbase + (int64_t)k0 * p.k_st, ko[0], ko[1], ko[2], ko[3], d);
        dma16x4g(vbase + (int64_t)k0 * p.v_st, vo[0], vo[1], vo[2], vo[3], d + TILE_B);
    };

    const int kx = l31 & 15;
    const int k_lane = l31 * 256 + (((kx & 14) | (lh ^ (kx & 1))) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int v_lane = (4 * lh + (sg >> 2)) * 256 + ((((sg >> 2) << 2) | (gh << 1) | ((sg >> 1) & 1)) << 4) + (sg & 1) * 8;
    // K fragment (32 keys of half h x 16 d at k-step ks) / V^T fragment (32 d of tile dt x 16 keys of step c of half h)
    auto kfrag = [&](int h, int ks) {
        const unsigned char* sk = smem + ((h >> 1) & 3) * STAGE_B + (h & 1) * 32 * 256;
        union { uint4 r; frag_t f; } u;
        u.r = *reinterpret_cast<const uint4*>(sk + (k_lane ^ (ks * 32)));
        return u.f;
    };
    auto vfrag = [&](int h, int c, int dt) {
        const int hc = h < 0 ? 0 : h;                                    // h = -1: the all-zero P of the prologue times tile 0's V
        const unsigned char* a0 = smem + ((hc >> 1) & 3) * STAGE_B + TILE_B + (((hc & 1) * 2 + c) * 16) * 256 + (v_lane ^ (dt << 6));
        union { s16x4_t hh[2]; frag_t f; } va;
        va.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a0);
        va.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * 256));
        return va.f;
    };

    o_zero();                                                            // O^T accumulators [dt * 2 + qb] = a[0:127]
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int nh = 2 * (min(qs + 63, T_ - 1) / KT + 1);                  // half tiles this wave multiplies (<= 2 nt)
    const int nt_w = nh >> 1;
    typedef union { uint32_t w[4]; frag_t f; } pfrag_t;                  // P^T operand: 8 keys x this lane's q

    auto needs = [&](const float (&mt)[2]) {
        const bool need = (mt[0] > m_run[0] + LAZY_RESCALE_LOG2) || (mt[1] > m_run[1] + LAZY_RESCALE_LOG2);
        return __builtin_amdgcn_ballot_w64(need) != 0;
    };
    // exp2(s c - m) for 4 consecutive scores of q block qb (one fma + one exp each), their sum, two packed P words
    auto probs4 = [&](const f32x16_t& st, int r0, float m_ref, float& ls, pfrag_t& pf) {
        float e[4];
#if UAMD_A64_DBG & 16
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = st[r0 + j];
#else
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r0 + j], p.scale_log2, -m_ref));
#endif
        ls += (e[0] + e[1]) + (e[2] + e[3]);
        pf.w[(r0 & 7) >> 1] = pack_pair2<T>(e[0], e[1]);
        pf.w[((r0 & 7) >> 1) + 1] = pack_pair2<T>(e[2], e[3]);
    };

    // ---- FAST segment (every half tile but the wave's last two)
    auto seg_fast = [&](int h, f32x16_t (&cur)[2], f32x16_t (&nxt)[2], pfrag_t (&pcur)[2][2], pfrag_t (&pprev)[2][2]) {
        // every K / V^T fragment of the segment is requested NOW (64 registers): with one wave per SIMD nobody hides an LDS
        // round trip, and a fragment requested one chunk ahead arrives after its MFMAs want to issue
        frag_t kfa[8], vfa[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kfa[ks] = kfrag(h + 1, ks);
#pragma unroll
        for (int j = 0; j < 8; ++j) vfa[j] = vfrag(h - 1, j >> 2, j & 3);
        __builtin_amdgcn_sched_barrier(0);
        // part A: row max of S(h)  ||  S(h+1), k-steps 0..3
        float mx[2] = {cur[0][0], cur[1][0]};
        // one MFMA, then a few VALU instructions, fenced: a wave issues in order, so MFMAs placed back to back make the
        // VALU work behind them wait for the matrix pipe (measured: MFMA time and VALU time simply added up)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (ks == 0) s_mfma_first<T>(nxt[qb], kfa[0], qf[qb][0]);
                else s_mfma<T>(nxt[qb], kfa[ks], qf[qb][ks]);
#pragma unroll
                for (int j = 0; j < 4; ++j) mx[qb] = fmaxf(mx[qb], cur[qb][4 * ks + j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float mt[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) mt[qb] = max_across_halves(mx[qb]) * p.scale_log2;
        if (__builtin_expect(needs(mt), 0)) {
            // rare: a row's max jumped. The pending product P(h-1) V(h-1) is relative to the old references: add it now,
            // zero P(h-1) (part B then adds nothing), raise the references, rescale l and O (asm on the hidden registers)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pv_i<T>((j & 3) * 2 + 0, vfa[j], pprev[j >> 2][0].f);
                pv_i<T>((j & 3) * 2 + 1, vfa[j], pprev[j >> 2][1].f);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pprev[c][qb].w[j] = 0u;
            float alpha[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float m_new = fmaxf(m_run[qb], mt[qb]);
                alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - (m_new == -INFINITY ? 0.f : m_new));
                m_run[qb] = m_new;
                l_run[qb] *= alpha[qb];
            }
            scale_pinned<0>(alpha[0]); scale_pinned<1>(alpha[1]); scale_pinned<2>(alpha[0]); scale_pinned<3>(alpha[1]);
            scale_pinned<4>(alpha[0]); scale_pinned<5>(alpha[1]); scale_pinned<6>(alpha[0]); scale_pinned<7>(alpha[1]);
        }
        // part B: P(h) = exp2(S(h) c - m)  ||  O += V(h-1) P(h-1) (16 MFMAs), S(h+1) k-steps 4..7 (8 MFMAs)
        const float m_ref[2] = {m_run[0], m_run[1]};                    // finite: the first half tile always takes the branch above
        float ls[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = j >> 2, dt = j & 3, qv = j >> 2, r0 = 4 * (j & 3);
            float e[4];
            pv_i<T>(dt * 2 + 0, vfa[j], pprev[c][0].f);
            e[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[qv][r0 + 0], p.scale_log2, -m_ref[qv]));
            e[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[qv][r0 + 1], p.scale_log2, -m_ref[qv]));
            __builtin_amdgcn_sched_barrier(0);
            pv_i<T>(dt * 2 + 1, vfa[j], pprev[c][1].f);
            e[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[qv][r0 + 2], p.scale_log2, -m_ref[qv]));
            e[3] = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[qv][r0 + 3], p.scale_log2, -m_ref[qv]));
            __builtin_amdgcn_sched_barrier(0);
            if (j >= 6) s_mfma_last<T>(nxt[j & 1], kfa[7], qf[j & 1][7]);
            else s_mfma<T>(nxt[j & 1], kfa[4 + (j >> 1)], qf[j & 1][4 + (j >> 1)]);
            ls[qv] += (e[0] + e[1]) + (e[2] + e[3]);
            pcur[r0 >> 3][qv].w[(r0 & 7) >> 1] = pack_pair2<T>(e[0], e[1]);
            pcur[r0 >> 3][qv].w[((r0 & 7) >> 1) + 1] = pack_pair2<T>(e[2], e[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run[0] += ls[0];
        l_run[1] += ls[1];
    };
    // ---- GENERAL segment: mask (the wave's diagonal / ragged tile), raise the references when needed, no interleaving
    auto seg_slow = [&](int h, f32x16_t (&cur)[2], f32x16_t (&nxt)[2], pfrag_t (&pcur)[2][2], const pfrag_t (&pprev)[2][2],
                        auto masked) {
        float mt[2];
        if (decltype(masked)::value) {
            const int k0 = h * 32;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key > q_pos[qb] || key >= T_) cur[qb][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float m = cur[qb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, cur[qb][r]);
            mt[qb] = max_across_halves(m) * p.scale_log2;
        }
        // the pending product first: it is relative to the old references
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const frag_t vf = vfrag(h - 1, c, dt);
                pv_i<T>(dt * 2 + 0, vf, pprev[c][0].f);
                pv_i<T>(dt * 2 + 1, vf, pprev[c][1].f);
            }
        if (needs(mt) || __builtin_amdgcn_ballot_w64(m_run[0] == -INFINITY || m_run[1] == -INFINITY) != 0) {
            float alpha[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float m_new = fmaxf(m_run[qb], mt[qb]);
                alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - (m_new == -INFINITY ? 0.f : m_new));
                m_run[qb] = m_new;
                l_run[qb] *= alpha[qb];
            }
            scale_pinned<0>(alpha[0]); scale_pinned<1>(alpha[1]); scale_pinned<2>(alpha[0]); scale_pinned<3>(alpha[1]);
            scale_pinned<4>(alpha[0]); scale_pinned<5>(alpha[1]); scale_pinned<6>(alpha[0]); scale_pinned<7>(alpha[1]);
        }
        // S(h+1): for the wave's very last half tile this multiplies whatever the next ring stage holds (never read)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const frag_t kf = kfrag(h + 1, ks);
            if (ks == 0) { s_mfma_first<T>(nxt[0], kf, qf[0][0]); s_mfma_first<T>(nxt[1], kf, qf[1][0]); }
            else if (ks == 7) { s_mfma_last<T>(nxt[0], kf, qf[0][7]); s_mfma_last<T>(nxt[1], kf, qf[1][7]); }
            else { s_mfma<T>(nxt[0], kf, qf[0][ks]); s_mfma<T>(nxt[1], kf, qf[1][ks]); }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float m_ref = m_run[qb] == -INFINITY ? 0.f : m_run[qb];
            float ls = 0.f;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 4) probs4(cur[qb], r0, m_ref, ls, pcur[r0 >> 3][qb]);
            l_run[qb] += ls;
        }
    };

    // ---- prologue: tiles 0, 1, 2 in flight; scores of half tile 0; an all-zero "previous P"
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    if (nt > 2) issue(2, 2);
    if (nt > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const frag_t kf = kfrag(0, ks);
        if (ks == 0) { s_mfma_first<T>(sa[0], kf, qf[0][0]); s_mfma_first<T>(sa[1], kf, qf[1][0]); }
        else if (ks == 7) { s_mfma_last<T>(sa[0], kf, qf[0][7]); s_mfma_last<T>(sa[1], kf, qf[1][7]); }
        else { s_mfma<T>(sa[0], kf, qf[0][ks]); s_mfma<T>(sa[1], kf, qf[1][ks]); }
    }
    // MID-trip step of trip t (between its two segments): segment 2t was the last reader of tile t-1, segment 2t+1 is
    // the first reader of tile t+1. Wait for tile t+1 (issued two trips ago; tile t+2 may still fly), barrier (every
    // wave is past segment 2t), refill the freed stage with tile t+3.
    auto ring_step = [&](int t) {
        if (t + 1 < nt) {
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 3 < nt) issue(t + 3, (t + 3) & 3);
        }
    };
    int t = 0;
    for (; t < nt_w - 1; ++t) {
        seg_fast(2 * t, sa, sb, pa, pb);
        ring_step(t);
        seg_fast(2 * t + 1, sb, sa, pb, pa);
    }
    seg_slow(2 * t, sa, sb, pa, pb, std::true_type{});     // t == nt_w - 1: the wave's diagonal (and maybe ragged) tile
    ring_step(t);
    seg_slow(2 * t + 1, sb, sa, pb, pa, std::true_type{});
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const frag_t vf = vfrag(nh - 1, c, dt);
            pv_i<T>(dt * 2 + 0, vf, pb[c][0].f);
            pv_i<T>(dt * 2 + 1, vf, pb[c][1].f);
        }
    for (int tt = nt_w; tt < nt; ++tt) ring_step(tt);    // G < 4: a wave with an earlier q subtile keeps the ring going
    // ---- epilogue: O = O^T / l, LSE = ln2 * (m + log2 l)
    auto store_o = [&](auto dt_c, auto qb_c) {
        constexpr int dt = decltype(dt_c)::value, qb = decltype(qb_c)::value;
        float f[16];
        read_pinned<dt * 2 + qb>(f);
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        if (q_pos[qb] < T_) {
            T* op = (T*)p.O + b * p.o_sb + (int64_t)q_pos[qb] * p.o_st + (int64_t)head * p.o_sh;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = dt * 32 + qd * 8 + lh * 4;
                uint2 ov;
                ov.x = pack_pair2<T>(f[qd * 4 + 0] * inv, f[qd * 4 + 1] * inv);
                ov.y = pack_pair2<T>(f[qd * 4 + 2] * inv, f[qd * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + d) = ov;
            }
            if (dt == 0 && lh == 0)
                p.LSE[((int64_t)b * p.Hq + head) * p.lse_st + q_pos[qb]] = (m_run[qb] + log2f(l_tot)) * 0.6931471805599453f;
        }
    };
    store_o(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    store_o(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    store_o(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    store_o(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    store_o(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    store_o(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
    store_o(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
    store_o(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
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
    int jt, pair_;
    block_to_work((int)blockIdx.x, (T_ + KT - 1) / KT, npairs, p.xcd_map, jt, pair_);
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



// ------------------------------------------------------------------------------------------------------------
// Backward, part 1, round-3 structure: dQ (+ Delta, LSE2) with ONE wave per SIMD -- attn_bwd_dkdv4_kernel with the roles of
// Q and K swapped. What was wrong with attn_bwd_dq_kernel (8 waves x 32 query rows, 246 us in the step = 33 % MFMA-busy):
// the compiler serialises `ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma` under its register pressure (16 waits per 16 MFMAs in
// the hot blocks) and every K / V / K^T fragment read from LDS feeds ONE MFMA.
// Here a block = 4 waves = the 4 query heads of a KV head (G = 8: two passes; G < 4: query-tile slices), each wave owns
// 64 query rows of its head and its SIMD's whole register file:
//   * dQ^T [128 d x 64 q] lives in AGPRs a0..a127 (tuples 4 kh + dt of attn_acc256.inc, asm-owned); the Q^T and dO^T
//     fragments of the wave's 64 rows (the B operands of S^T and dP^T) are parked in a128..a255 for the whole pass (MFMA
//     source operands may come from the accumulator file): no register pressure, no LDS traffic for them. First version:
//     Q^T in 64 VGPRs + dO read from an LDS tile -- 4 scratch reloads in every step's DMA preamble (the knock-out
//     builds showed ~120 us of the kernel to be independent of MFMAs, VALU and LDS reads alike: profiles/r03o_dq4_knockout.jsonl);
//   * K / V arrive in tiles of 32 keys through a double-buffered LDS ring SHARED by the four waves (each wave DMAs a
//     quarter of a tile): one `vmcnt(0) + s_barrier` per step -- the four heads have identical masks, so they stay in step;
//   * every K / V / K^T fragment read feeds TWO MFMAs (the two query halves); a step is a hand-pipelined stream of 24 chunks
//     of two v_mfma_f32_32x32x16:  S^T = K Q^T (8) | dP^T = V dO^T (8, P = exp2(S^T c - LSE2[q]) beside them) |
//     dQ^T += K^T dS' (8), dS' = P (dP^T - Delta[q]) -- the first half of dS' sits between the phases (dP must be complete), the
//     second half beside the first dQ chunks; the softmax scale is applied once in the epilogue;
//   * LSE2 and Delta are per-LANE scalars here (lane = query row): no statistics traffic at all.
// Also writes Delta (plane 0) and LSE2 = LSE log2(e) (plane 1) for attn_bwd_dkdv4_kernel, like the old kernel.
// LDS: only the ring, stage x (K tile 8 KiB | V tile 8 KiB) = 32 KiB.
constexpr int DQ4_LDS = 2 * 16384;
// -DUAMD_DQ4_KO=bits: knock-out builds for timing only (results are garbage): 1 = no softmax / dS arithmetic, 2 = operand
// fragments are read from LDS once per step instead of once per chunk, 4 = no per-step barrier / DMA, 8 = no MFMAs
#ifndef UAMD_DQ4_KO
#define UAMD_DQ4_KO 0
#endif
#ifndef UAMD_DQ4_PF
#define UAMD_DQ4_PF 2             // operand prefetch distance in chunks
#endif

template <typename T, bool BAND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) attn_bwd_dq4_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename MfmaA<T>::frag frag_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int unit = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = p.G, T_ = p.T;
    const int hpp = G < 4 ? G : 4;                    // heads per pass
    const int npass = G / hpp, nslice = 4 / hpp;
    const int hin = unit % hpp, slice = unit / hpp;
    const int npairs = p.Hk * p.B;
    const int nq64 = (T_ + 63) / 64, nqb = (nq64 + nslice - 1) / nslice;
    int rank_, pair_;
    block_to_work((int)blockIdx.x, nqb, npairs, p.xcd_map, rank_, pair_);
    const int jb = nqb - 1 - rank_;                                     // the last query tiles see every key: heaviest first
    const int kvh = pair_ % p.Hk, b = pair_ / p.Hk;
    const int q0 = (jb * nslice + slice) * 64;                          // this wave's query tile (>= T: an idle wave)
    const bool active = q0 < T_;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u8*)smem;
    int qrow[2], q_ld[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        qrow[kh] = q0 + kh * 32 + l31;
        q_ld[kh] = qrow[kh] < T_ ? qrow[kh] : T_ - 1;
    }
    // key-tile (32 keys) range of the block and of this wave
    const int q0_first = min(jb * nslice * 64, T_ - 1), q0_last = min((jb * nslice + nslice - 1) * 64 + 63, T_ - 1);
    const int t_first_blk = BAND ? p.lo[(int64_t)b * T_ + q0_first] / 32 : 0;
    const int nsteps = q0_last / 32 - t_first_blk + 1;
    const int t_first_w = BAND ? p.lo[(int64_t)b * T_ + min(q0, T_ - 1)] / 32 : 0;
    const int t_last_w = min(q0 + 63, T_ - 1) / 32;
    int lo_q[2] = {0, 0};
    if (BAND) {
        lo_q[0] = p.lo[(int64_t)b * T_ + q_ld[0]];
        lo_q[1] = p.lo[(int64_t)b * T_ + q_ld[1]];
    }
    const int lo_max_w = BAND ? p.lo[(int64_t)b * T_ + min(q0 + 63, T_ - 1)] : 0;   // (lo is non-decreasing in q)

    // ---- DMA plumbing. Piece i of a 16-row group: rows 4 i + (lane >> 4), source slot (lane & 15) ^ swz_c(row)
    const int dsw0 = ((lane & 15) ^ ((lane >> 4) << 2)) << 4;
    // ring: this wave fetches part `unit` of every K | V tile: unit 0 / 1 = K rows 0-15 / 16-31, 2 / 3 = V rows 0-15 / 16-31
    const bool my_v = unit >= 2;
    const int64_t t_st = my_v ? p.v_st : p.k_st;
    const T* t_base = my_v ? (const T*)p.V + b * p.v_sb + (int64_t)kvh * p.v_sh : (const T*)p.K + b * p.k_sb + (int64_t)kvh * p.k_sh;
    const int t_r0 = (unit & 1) * 16;                 // first tile row of this wave's part
    unsigned to_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        to_[i] = (unsigned)((int64_t)(t_r0 + i * 4 + (lane >> 4)) * t_st * 2) + (unsigned)(dsw0 ^ (i << 4));
    auto issue = [&](int t, int stage) {
        const int k0 = t * 32;
        const unsigned d = lds_base + stage * 16384 + unit * 4096;
        if (k0 + 32 <= T_) {
            dma16x4g(t_base + (int64_t)k0 * t_st, to_[0], to_[1], to_[2], to_[3], d);
        } else {                                                        // ragged last tile: rows past the end re-read the last row
            unsigned o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = min(t_r0 + i * 4 + (lane >> 4), T_ - 1 - k0);
                o[i] = (unsigned)((int64_t)r * t_st * 2) + (unsigned)(dsw0 ^ (i << 4));
            }
            dma16x4g(t_base + (int64_t)k0 * t_st, o[0], o[1], o[2], o[3], d);
        }
    };

    // ---- per-lane ABSOLUTE LDS byte addresses (swizzle C), made opaque once: stage / operand / k-step are immediates
    const int r_lane = l31 * 256 + ((swz_c(l31 & 15) ^ lh) << 4);
    const int sg = lane & 15, gh = (lane >> 4) & 1;
    const int t_lane = (4 * lh + (sg >> 2)) * 256 +
                       ((((sg >> 2) << 2) | (((gh << 1) | ((sg >> 1) & 1)) ^ lh)) << 4) + (sg & 1) * 8;
    unsigned ck[8], ct[4], ct2[4];                     // row reads / transposing reads of the ring
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        ck[ks] = lds_base + (unsigned)(r_lane ^ (ks * 32));
        asm volatile("" : "+v"(ck[ks]));
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        ct[dt] = lds_base + (unsigned)(t_lane ^ (dt << 6));
        ct2[dt] = lds_base + (unsigned)((t_lane ^ (dt << 6)) ^ 32) + 8 * 256;
        asm volatile("" : "+v"(ct[dt]), "+v"(ct2[dt]));
    }

    for (int pass = 0; pass < npass; ++pass) {
        const int head = kvh * G + pass * hpp + hin;
        // ---- Q^T and dO^T fragments -> a128..a255 (fragment kh * 8 + ks and 16 + kh * 8 + ks), Delta / LSE2 per lane
        acc256_zero();
        float lse2[2], delta[2];
        static_for<2>([&](auto khc) {
            constexpr int kh = decltype(khc)::value;
            const T* qp = (const T*)p.Q + b * p.q_sb + (int64_t)q_ld[kh] * p.q_st + (int64_t)head * p.q_sh + lh * 8;
            const T* dp_ = (const T*)p.dO + b * p.do_sb + (int64_t)q_ld[kh] * p.do_st + (int64_t)head * p.do_sh + lh * 8;
            const T* op = (const T*)p.O + b * p.o_sb + (int64_t)q_ld[kh] * p.o_st + (int64_t)head * p.o_sh + lh * 8;
            float dl = 0.f;
            static_for<8>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                union { uint4 r; T e[8]; } u, d, o;
                u.r = *reinterpret_cast<const uint4*>(qp + ks * 16);
                d.r = *reinterpret_cast<const uint4*>(dp_ + ks * 16);
                o.r = *reinterpret_cast<const uint4*>(op + ks * 16);
                acc256_bset<kh * 8 + ks>(u.r.x, u.r.y, u.r.z, u.r.w);
                acc256_bset<16 + kh * 8 + ks>(d.r.x, d.r.y, d.r.z, d.r.w);
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += to_f32(d.e[j]) * to_f32(o.e[j]);
            });
            dl += __shfl_xor(dl, 32, 64);
            const int64_t stat_idx = ((int64_t)b * p.Hq + head) * p.lse_st + q_ld[kh];
            lse2[kh] = p.LSE[stat_idx] * 1.4426950408889634f;
            delta[kh] = dl;
            if (lh == 0 && qrow[kh] < T_) {
                p.Delta[stat_idx] = dl;
                p.Delta[(int64_t)p.B * p.Hq * p.lse_st + stat_idx] = lse2[kh];
            }
        });

        auto body = [&](auto masked, auto stage_c, int k0) {
            constexpr bool MASK = decltype(masked)::value;
            constexpr int STAGE = decltype(stage_c)::value;
            constexpr int SO = STAGE * 16384;
            f32x16_t sc[2], dp[2];
            union { uint32_t w[8]; frag_t f[2]; } sb[2];          // dS' as B operands: f[c] = keys 16 c .. 16 c + 15
            constexpr int PF = UAMD_DQ4_PF;
            frag_t ob[PF + 1][1];
            auto rd128 = [&](unsigned addr) {
                union { u32x4a_t r; frag_t f; } u;
                u.r = *(const lds_u32x4a*)(uintptr_t)addr;
                return u.f;
            };
            auto rdtr = [&](unsigned a0, unsigned a1) {
                union { s16x4_t h[2]; frag_t f; } t;
                t.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a0);
                t.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a1);
                return t.f;
            };
            auto reads = [&](auto kc, frag_t* o) {
                constexpr int k = decltype(kc)::value;
                if constexpr ((UAMD_DQ4_KO & 2) != 0 && (k & 7) >= 3) {
                    o[0] = ob[0][0];
                } else if constexpr (k < 8) {
                    o[0] = rd128(ck[k] + SO);                                 // K rows
                } else if constexpr (k < 16) {
                    o[0] = rd128(ck[k - 8] + (SO + 8192));                    // V rows
                } else {
                    constexpr int c = (k - 16) >> 2, dt = (k - 16) & 3;
                    o[0] = rdtr(ct[dt] + (SO + c * 4096), ct2[dt] + (SO + c * 4096));   // K^T
                }
            };
            auto mfmas = [&](auto kc, const frag_t* o, auto half_c) {        // half = query half
                constexpr int k = decltype(kc)::value, kh = decltype(half_c)::value;
                if constexpr ((UAMD_DQ4_KO & 8) != 0) {
                    asm volatile("" :: "v"(o[0]));
                } else if constexpr (k < 8) {
                    acc256_vmfma_b<T, kh * 8 + k, k == 0>(sc[kh], o[0]);                    // x Q^T fragment (kh, ks = k)
                } else if constexpr (k < 16) {
                    acc256_vmfma_b<T, 16 + kh * 8 + (k - 8), k == 8>(dp[kh], o[0]);         // x dO^T fragment (kh, ks = k - 8)
                } else {
                    constexpr int c = (k - 16) >> 2, dt = (k - 16) & 3;
                    acc256_mfma<T, 4 * kh + dt>(o[0], sb[kh].f[c]);
                }
            };
            // pair pi = 2 j + kh: registers 2 j, 2 j + 1 of half kh = keys k0 + 8 (j >> 1) + 4 lh + 2 (j & 1) + {0, 1}
            auto p_pair = [&](auto pic) {
                constexpr int pi = decltype(pic)::value, kh = pi & 1, j = pi >> 1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kh][2 * j + e], p.scale_log2, -lse2[kh]));
                    if (MASK) {
                        const int r = 2 * j + e;
                        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key > qrow[kh] || key >= T_ || qrow[kh] >= T_ || (BAND && key < lo_q[kh])) pv = 0.f;
                    }
                    sc[kh][2 * j + e] = pv;
                }
                return sc[kh][2 * j] + sc[kh][2 * j + 1];
            };
            auto ds_pair = [&](auto pic) {                        // masked entries have P = 0, hence dS' = 0
                constexpr int pi = decltype(pic)::value, kh = pi & 1, j = pi >> 1;
                const float x0 = sc[kh][2 * j] * (dp[kh][2 * j] - delta[kh]);
                const float x1 = sc[kh][2 * j + 1] * (dp[kh][2 * j + 1] - delta[kh]);
                sb[kh].w[j] = pack_pair2<T>(x0, x1);
                return sb[kh].w[j];
            };
            // VALU work of chunk k: P pairs in chunks 9-15 (16 pairs: 3 3 2 2 2 2 2, one in the chunk's first half), the
            // second half of the dS' pairs (8-15) in chunks 16-19 (two per chunk, one per half)
            auto valu = [&](auto kc, auto half_c) {
                constexpr int k = decltype(kc)::value;
                constexpr bool FIRST = decltype(half_c)::value == 0;
                if constexpr ((UAMD_DQ4_KO & 1) != 0) {
                    if constexpr (k == 16 && FIRST) { sb[0].f[1] = ob[0][0]; sb[1].f[1] = ob[0][0]; }
                } else if constexpr (k >= 9 && k < 16) {
                    constexpr int slot = k - 9, first = slot < 2 ? 3 * slot : 6 + 2 * (slot - 2), count = slot < 2 ? 3 : 2;
                    if constexpr (FIRST) {
                        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));     // S is complete: pin the reads of it BEHIND this point
                        const float w0 = p_pair(std::integral_constant<int, first>{});
                        asm volatile("" :: "v"(w0));
                    } else {
                        const float w1 = p_pair(std::integral_constant<int, first + 1>{});
                        float w2 = w1;
                        if constexpr (count == 3) w2 = p_pair(std::integral_constant<int, first + 2>{});
                        asm volatile("" :: "v"(w1), "v"(w2));
                    }
                } else if constexpr (k >= 16 && k < 20) {
                    constexpr int first = 8 + 2 * (k - 16);
                    if constexpr (FIRST) {
                        const uint32_t w0 = ds_pair(std::integral_constant<int, first>{});
                        asm volatile("" :: "v"(w0));
                    } else {
                        const uint32_t w1 = ds_pair(std::integral_constant<int, first + 1>{});
                        asm volatile("" :: "v"(w1));
                    }
                }
            };
            static_for<PF>([&](auto kc) { reads(kc, ob[decltype(kc)::value]); });
            static_for<24>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k + PF < 24) reads(std::integral_constant<int, k + PF>{}, ob[(k + PF) % (PF + 1)]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (k == 16 && (UAMD_DQ4_KO & 1) != 0) {
                    sb[0].f[0] = ob[0][0]; sb[1].f[0] = ob[0][0];
                }
                if constexpr (k == 16 && (UAMD_DQ4_KO & 1) == 0) {
                    // dP is complete only now: the first half of dS' (pairs 0-7 = the c = 0 operands of both query halves)
                    // cannot ride beside an MFMA that does not need it. The asm MFMAs are invisible to the hazard
                    // recognizer: the XDL write -> VALU read wait states are spelled out.
                    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                    asm volatile("" : "+v"(dp[0]), "+v"(dp[1]));
                    uint32_t w[8];
                    static_for<8>([&](auto pc) { w[decltype(pc)::value] = ds_pair(pc); });
                    asm volatile("s_nop 3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfmas(kc, ob[k % (PF + 1)], std::integral_constant<int, 0>{});
                valu(kc, std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kc, ob[k % (PF + 1)], std::integral_constant<int, 1>{});
                valu(kc, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto run = [&](auto stage_c, int step) {
            constexpr int STAGE = decltype(stage_c)::value;
            if ((UAMD_DQ4_KO & 4) == 0 || step == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's piece of the tile landed
            __builtin_amdgcn_s_barrier();                          // ... everybody's did; everybody is done with the other stage
            asm volatile("" ::: "memory");
            // the next tile into the other stage -- last step: this tile once more (unconditional: no branch around a DMA)
            issue(t_first_blk + (step + 1 < nsteps ? step + 1 : step), STAGE ^ 1);
            }
            const int t = t_first_blk + step, k0 = t * 32;
            if (!active || t < t_first_w || t > t_last_w) return;
            const bool slow = (k0 + 31 > q0) || (k0 + 32 > T_) || (q0 + 64 > T_) || (BAND && k0 < lo_max_w);
            if (slow) body(std::true_type{}, stage_c, k0); else body(std::false_type{}, stage_c, k0);
        };
        if (nsteps > 0) issue(t_first_blk, 0);
        for (int step = 0; step < nsteps; step += 2) {
            run(std::integral_constant<int, 0>{}, step);
            if (step + 1 < nsteps) run(std::integral_constant<int, 1>{}, step + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last step's spare DMA has landed
        // ---- dQ = scale * dQ^T: tuple 4 kh + dt = d rows dt * 32 + (r & 3) + 8 (r >> 2) + 4 lh of query qrow[kh]
        auto store_dq = [&](auto ic) {
            constexpr int I = decltype(ic)::value, kh = I >> 2, dt = I & 3;
            float f[16];
            acc256_read<I>(f);
            if (active && qrow[kh] < T_) {
                T* op = (T*)p.dQ + b * p.dq_sb + (int64_t)qrow[kh] * p.dq_st + (int64_t)head * p.dq_sh;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int d = dt * 32 + qd * 8 + lh * 4;
                    uint2 o;
                    o.x = pack_pair2<T>(f[4 * qd + 0] * p.scale, f[4 * qd + 1] * p.scale);
                    o.y = pack_pair2<T>(f[4 * qd + 2] * p.scale, f[4 * qd + 3] * p.scale);
                    *reinterpret_cast<uint2*>(op + d) = o;
                }
            }
        };
        static_for<8>(store_dq);
        if (pass + 1 < npass) __syncthreads();                    // the ring is re-filled from the first key tile
    }
}

