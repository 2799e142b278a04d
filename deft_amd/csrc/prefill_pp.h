// Causal prefill, ping-pong form (round 4): prefill_kernel's arithmetic and data path with the two waves of every SIMD HALF A TILE
// OUT OF PHASE, so that one's MFMA segment always runs beside the other's VALU segment.
//
// Why.  Per 128-key tile a wave issues 32 QK^T MFMAs (1024 cycles of its SIMD's matrix pipe), 64 exp2 with their fma / max /
// convert / add (~2000 cycles of VALU issue at 4 cycles per wave64 instruction), 32 PV MFMAs (1024).  The two waves a SIMD holds
// (w and w + 4 of the 512-thread workgroup) therefore need 4096 cycles of matrix pipe and ~4000 of VALU per tile -- which
// overlap only if the waves are in DIFFERENT segments.  prefill_kernel runs all eight waves in phase (one barrier per tile): both
// waves of a SIMD want the matrix pipe at the same time, then the VALU at the same time, and a tile takes ~10 000 cycles (4.5 us:
// 0.36-0.41 of the MFMA peak).  Round 1 shifted waves 4-7 by one PV segment (PV(t-1) | QK^T(t) | softmax(t) against QK^T(t) |
// softmax(t) | PV(t)): one of the three slots then has MFMA beside MFMA and the softmax is one long slot; neutral.
//
// Here a tile is FOUR slots -- Q (QK^T), S1 (row maxima, rescale, first half of the exponentials), S2 (second half, row sums),
// P (PV) -- each ended by a workgroup barrier, and waves 4-7 run two slots behind waves 0-3:
//
//      slot        4p        4p+1       4p+2       4p+3
//      waves 0-3   Q(p)      S1(p)      S2(p)      P(p)
//      waves 4-7   S2(p-1)   P(p-1)     Q(p)       S1(p)          every slot: one MFMA segment beside one VALU segment
//
// K / V tiles double-buffered in LDS as before, K and V requested separately: K(p+1) at slot 4p (its buffer was last read by
// waves 4-7's Q(p-1) in slot 4p-2), V(p+1) at slot 4p+2 (last read by their P(p-1) in slot 4p+1); every wave waits for its own
// pieces (counted) in front of the barrier behind which they are first read.
//
// Included by deft_kernels.hip after prefill.h (PrefillParams, PrefillSmem).
#pragma once

namespace deft {

template <int D>
__global__ __launch_bounds__(512, 1) void prefill_pp_kernel(PrefillParams p) {
    constexpr int KS = D / 16;
    constexpr int QB = 256;
    static_assert(D == 128, "prefill is instantiated for head_dim 128");
    using SM = PrefillSmem<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2;  // 0: waves 0-3, 1: waves 4-7 (two slots behind)
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    // (grid order: prefill_kernel's -- longest blocks first across all heads and sequences, a KV head's q heads on one XCD)
    const int per_row = p.Hq * p.batch;
    const int L = (int)blockIdx.x;
    const int m = p.nblk - 1 - L / per_row;
    const int rem = L - (L / per_row) * per_row;
    const int b = rem / p.Hq;
    const int hi = rem - b * p.Hq;
    const int Hkv = p.Hq / p.G;
    const int head = (hi % Hkv) * p.G + hi / Hkv;
    const int len = p.b_seq_len[b];
    const int64_t start = p.b_start_loc[b];
    if (m * QB >= len) return;
    const int kvh = head / p.G;
    // the second-dispatched half loses VALU arbitration to the older one on every segment (MI355X_MICROARCH.md, "Two waves per
    // SIMD", item 4): one static priority for it, no per-segment flips
    if (grp) __builtin_amdgcn_s_setprio(1);

    const int dpos = l & 15, dkey = l >> 4;
    const int tg = l >> 4, tx = l & 15;
    int vtr_col_b[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) vtr_col_b[blk] = (4 * (blk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
    const int vtr_row_b = (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    const int krow_b = c * D * 2;
    const int kcol_b = ((h ^ c) & 15) * 16;
    int kfrag_b[KS], vfrag_b[4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfrag_b[ks] = krow_b + (kcol_b ^ (32 * ks));
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) vfrag_b[blk] = vtr_row_b + vtr_col_b[blk];

    const char* kbase = reinterpret_cast<const char*>(p.k + start * p.k_st + (int64_t)kvh * p.k_sh);
    const char* vbase = reinterpret_cast<const char*>(p.v + start * p.v_st + (int64_t)kvh * p.v_sh);
    // 4 K (or 4 V) instructions per wave: keys 16w + 4i + dkey of tile t
    auto issue_k = [&](int t, int stg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = 16 * w + 4 * i + dkey;
            int tok = TILE * t + key;
            tok = tok < len ? tok : len - 1;  // padding aliases the last token; masked by the causal test
            const int kc = (dpos ^ (key & 15)) * 16;
            dma16(kbase + (int64_t)tok * p.k_st * 2 + kc, SM::K_OFF + (uint32_t)stg * SM::STAGE + (uint32_t)(16 * w + 4 * i) * 256u);
        }
    };
    auto issue_v = [&](int t, int stg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = 16 * w + 4 * i + dkey;
            int tok = TILE * t + key;
            tok = tok < len ? tok : len - 1;
            const int vc = (dpos ^ (4 * (key & 3))) * 16;
            dma16(vbase + (int64_t)tok * p.v_st * 2 + vc, SM::V_OFF + (uint32_t)stg * SM::STAGE + (uint32_t)(16 * w + 4 * i) * 256u);
        }
    };

    const int ntiles = min(2 * m + 2, (len + TILE - 1) / TILE);
    issue_k(0, 0);
    issue_v(0, 0);

    const int qi = m * QB + 32 * w + c;
    const int qrow = qi < len ? qi : len - 1;
    half8 qf[KS];
    {
        const _Float16* qp = p.q + (start + qrow) * p.q_st + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qp + 16 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[bk][r] = 0.f;
    const int q_lo = m * QB + 32 * w;

    // ---- a tile's four segments (state between them: acc from Q to S2, msafe / sum from S1 to S2, pb from S1 / S2 to P) --------
    floatx16 acc[4];
    half8 pb[4][2];
    float msafe = 0.f, alpha = 1.f, sum = 0.f;
    auto live = [&](int t) { return TILE * t <= q_lo + 31; };  // (wave-uniform) some key of tile t is at or below some query of the wave

    auto seg_q = [&](int t) {
        if (!live(t)) return;
        const int stg = t & 1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
        const int kstage = SM::K_OFF + stg * SM::STAGE;
        half8 af[2][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) af[0][kb] = *reinterpret_cast<const half8*>(smem + (kfrag_b[0] + kstage) + 32 * kb * D * 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
                const char* kp = smem + (kfrag_b[ks + 1] + kstage);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) af[(ks + 1) & 1][kb] = *reinterpret_cast<const half8*>(kp + 32 * kb * D * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][kb], qf[ks], acc[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto exp_half = [&](int kb0) {  // exponentials of key blocks kb0, kb0 + 1; row-sum chain continued
#pragma unroll
        for (int kb = kb0; kb < kb0 + 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[kb][r], p.scale_log2e, -msafe));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[kb][r + 1], p.scale_log2e, -msafe));
                pb[kb][r >> 3][r & 7] = (_Float16)e0;
                pb[kb][r >> 3][(r & 7) + 1] = (_Float16)e1;
                sum += e0;
                sum += e1;
            }
    };
    auto seg_s1 = [&](int t) {
        if (!live(t)) return;
        const int key0 = TILE * t;
        const bool diag = key0 + TILE - 1 > q_lo;
        float mx = -INFINITY;
        if (diag) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
                    acc[kb][r] = key <= qi ? acc[kb][r] : -INFINITY;
                    mx = fmaxf(mx, acc[kb][r]);
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[kb][r]);
        }
        mx = max_xor32(mx) * p.scale_log2e;
        const float m_new = fmaxf(m_run, mx);
        msafe = (m_new == -INFINITY) ? 0.f : m_new;
        alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
        m_run = m_new;
        sum = 0.f;
        exp_half(0);
    };
    auto seg_s2 = [&](int t) {
        if (!live(t)) return;
        exp_half(2);
        sum = sum_xor32(sum);
        l_run = l_run * alpha + sum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[bk][r] *= alpha;
        }
    };
    auto seg_p = [&](int t) {
        if (!live(t)) return;
        const int stg = t & 1;
        const int vstage = SM::V_OFF + stg * SM::STAGE;
        int vfrag[4];
#pragma unroll
        for (int bk = 0; bk < 4; ++bk) vfrag[bk] = vfrag_b[bk] + vstage;
        typedef __attribute__((address_space(3))) short4v* lds_s4;
        union VFrag {
            short4v s4[2];
            half8 h8;
        };
        VFrag vf[2][4];
        auto load_group = [&](int g, VFrag (&dst)[4]) {
            const int kb = g >> 1, tt = g & 1;
#pragma unroll
            for (int bk = 0; bk < 4; ++bk) {
                const int vb = vfrag[bk] + (32 * kb * D * 2 + (16 * tt) * D * 2);
                dst[bk].s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                dst[bk].s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
            }
        };
        load_group(0, vf[0]);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) load_group(g + 1, vf[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bk = 0; bk < 4; ++bk)
                o[bk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[g & 1][bk].h8, pb[g >> 1][g & 1], o[bk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    wait_vm<0>();
    lds_barrier();  // tile 0 complete
    // ---- periods: 4 slots each; waves 4-7 two slots behind (one extra period drains them).  The two groups run their own copy
    //      of the loop -- the same barriers and the same requests at the same places, different segments in between (one loop with
    //      a branch per slot made the register allocator carry both groups' live ranges everywhere: 426 spilled registers) -------
    // per slot, for every wave: slot 0 requests K(pd+1) [its buffer's last reader: waves 4-7's Q(pd-1), two slots ago]; the end of
    // slot 1 waits for this wave's pieces of V(pd) [younger: K(pd+1)]; slot 2 requests V(pd+1) [last reader: waves 4-7's P(pd-1),
    // the slot before]; the end of slot 3 waits for K(pd+1) [younger: V(pd+1)]
    if (grp == 0) {
        for (int pd = 0; pd <= ntiles; ++pd) {
            const bool mine = pd < ntiles, next = pd + 1 < ntiles;
            if (next) issue_k(pd + 1, (pd + 1) & 1);
            if (mine) seg_q(pd);
            lds_barrier();
            if (mine) seg_s1(pd);
            if (next) wait_vm<4>();
            else wait_vm<0>();
            lds_barrier();
            if (next) issue_v(pd + 1, (pd + 1) & 1);
            if (mine) seg_s2(pd);
            lds_barrier();
            if (mine) seg_p(pd);
            if (next) wait_vm<4>();
            lds_barrier();
        }
    } else {
        for (int pd = 0; pd <= ntiles; ++pd) {
            const bool mine = pd < ntiles, next = pd + 1 < ntiles;
            if (next) issue_k(pd + 1, (pd + 1) & 1);
            if (pd >= 1) seg_s2(pd - 1);
            lds_barrier();
            if (pd >= 1) seg_p(pd - 1);
            if (next) wait_vm<4>();
            else wait_vm<0>();
            lds_barrier();
            if (next) issue_v(pd + 1, (pd + 1) & 1);
            if (mine) seg_q(pd);
            lds_barrier();
            if (mine) seg_s1(pd);
            if (next) wait_vm<4>();
            lds_barrier();
        }
    }
    // ---- normalise and store: lane (c, h) holds d = 32 bk + 8 j + 4 h + (0..3) of query c ---------------------
    if (qi < len) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        _Float16* op = p.o + (start + qi) * p.o_st + (int64_t)head * p.o_sh + 4 * h;
#pragma unroll
        for (int bk = 0; bk < 4; ++bk)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half4 v4 = {(_Float16)(o[bk][4 * j] * inv), (_Float16)(o[bk][4 * j + 1] * inv),
                            (_Float16)(o[bk][4 * j + 2] * inv), (_Float16)(o[bk][4 * j + 3] * inv)};
                *reinterpret_cast<half4*>(op + 32 * bk + 8 * j) = v4;
            }
    }
}

}  // namespace deft
