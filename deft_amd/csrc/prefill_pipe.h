// Causal prefill attention, software-pipelined form: QK^T of the NEXT 64-key half-tile is issued between the softmax
// groups of the CURRENT one, so that the matrix pipe runs in the shadow of the wave's own VALU work.
//
// Included by deft_kernels.hip after prefill.h (PrefillParams, the grid order, the arithmetic and the LDS row formats are
// that kernel's; see its header for the reference interface: context_flashattention_nopad.py:130-195 behind
// DeFTAttention.prefill_forward_triton, deft_attention.py:50-70).
//
// Why: per-workgroup timelines of prefill_kernel (tools/prefill_timeline.py, 4k tokens) show 4.15 us per 128-key tile and
// workgroup against ~1.9 us of MFMA time (2 waves x 64 MFMAs x 32 cycles per SIMD): QK^T (MFMA), softmax (VALU), PV (MFMA)
// run one after the other in every wave, and the two waves of a SIMD -- in lockstep behind the per-tile barrier -- are
// in the same phase at the same time, so one pipe idles while the other works.  Here a wave's instruction stream
// alternates them itself:
//
//   * stages are 64 keys (16 KB of K, 16 KB of V), a ring of four each (128 KB of LDS, as before); one barrier per
//     half-tile; the DMA of half u + 3 is issued at the top of step u (its slot was last read in steps u - 2 / u - 1);
//   * step u, phase A: S(u + 1) = K(u + 1) Q^T -- 16 MFMAs, two per k-step -- with one eighth of softmax(S(u)) after every
//     pair (mask / max, max, scale + alpha, 4 x {exp2, cvt, dot2}, row sum); the K fragments of k-step ks + 1 are read
//     while the MFMAs of ks run;
//   * phase B: O *= alpha (when some row's maximum moved), O^T += V(u)^T P(u)^T -- 16 MFMAs;
//   * two score sets (S(u), S(u + 1): 32 + 32 registers) alternate by name (the step body is instantiated twice).
#pragma once

namespace deft {

template <int D>
struct PrefillPipeSmem {
    static constexpr int HALF = 64;             // keys per stage
    static constexpr int STAGE = HALF * D * 2;  // one K (or V) half-tile
    static constexpr int NST = 4;               // ring depth
    static constexpr int K_OFF = 0;
    static constexpr int V_OFF = NST * STAGE;
    static constexpr int BYTES = 2 * NST * STAGE;  // 128 KB
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

template <int D>
__global__ __launch_bounds__(512, 1) void prefill_pipe_kernel(PrefillParams p) {
    constexpr int KS = D / 16;
    constexpr int QB = 256;  // queries per workgroup
    constexpr int HALF = 64;
    static_assert(D == 128, "prefill is instantiated for head_dim 128");
    using SM = PrefillPipeSmem<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    // grid order: prefill.h (one linear grid, longest blocks first over all heads and sequences, KV head -> XCD)
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

    // ---- lane constants (LDS row formats of stage1_np.h: K chunks XOR-ed by key & 15, V chunks by 4 * (key & 3)) -----
    const int dpos = l & 15, dkey = l >> 4;
    const int tg = l >> 4, tx = l & 15;
    const int vtr_row_b = (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    int vfrag_b[4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) vfrag_b[bk] = vtr_row_b + (4 * (bk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
    // K fragment of k-step ks: row c of the key block, 16-byte chunk (2 ks + h) ^ (c & 15) -- the lane part of the XOR kept in
    // two registers, the k-step part applied per use (one v_xor + one v_add: eight stored bases were eight registers the
    // tile loop does not have)
    const int kbase_b = c * D * 2 + (((h ^ c) & 1) * 16);
    const int khi_b = ((h ^ c) & 14) * 16;
    auto kfrag = [&](int ks) { return kbase_b + (khi_b ^ (32 * ks)); };

    const char* kbase = reinterpret_cast<const char*>(p.k + start * p.k_st + (int64_t)kvh * p.k_sh);
    const char* vbase = reinterpret_cast<const char*>(p.v + start * p.v_st + (int64_t)kvh * p.v_sh);
    // Steps come in pairs (the two score sets alternate by name), so the count is rounded up to even: a half beyond the
    // sequence re-reads the last real one and is masked entirely (every key of it is > every valid query).
    const int nhalves = min(4 * m + 4, (len + HALF - 1) / HALF);
    const int nsteps = (nhalves + 1) & ~1;
    // 2 K + 2 V instructions per wave and half: keys 8 w + 4 i + dkey of half x, into slot x & 3.  The address is a scalar
    // base (the half's first token) + a 32-bit per-lane offset recomputed here (a few VALU operations) -- nothing per lane is
    // kept across steps: 64-bit per-lane addresses were what the register allocator spilled first, and a scratch reload in
    // front of every DMA costs a memory round trip per step.
    const uint32_t kS = (uint32_t)(p.k_st * 2), vS = (uint32_t)(p.v_st * 2);  // row strides in bytes (host-checked < 2^31 / 64)
    auto issue_half = [&](int x) {
        const uint32_t slot = (uint32_t)(x & (SM::NST - 1)) * SM::STAGE;
        const int xs = x < nhalves ? x : nhalves - 1;  // (the padding half of an odd count)
        const char* kh = kbase + (int64_t)HALF * xs * p.k_st * 2;  // scalar
        const char* vh = vbase + (int64_t)HALF * xs * p.v_st * 2;
        const int last = len - 1 - HALF * xs;  // padding aliases the sequence's last token (masked by the causal test)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = 8 * w + 4 * i + dkey;
            const uint32_t kc = (uint32_t)(key < last ? key : last);
            dma16s(kh, kc * kS + (uint32_t)((dpos ^ (key & 15)) * 16), SM::K_OFF + slot + (uint32_t)(8 * w + 4 * i) * 256u);
            dma16s(vh, kc * vS + (uint32_t)((dpos ^ (4 * (key & 3))) * 16), SM::V_OFF + slot + (uint32_t)(8 * w + 4 * i) * 256u);
        }
    };

    issue_half(0);
    if (nsteps > 1) issue_half(1);
    if (nsteps > 2) issue_half(2);

    // ---- this lane's query and its Q fragments (B operand: 8 halves at d = 16 ks + 8 h) ----------------------
    const int qi = m * QB + 32 * w + c;  // query index inside the sequence
    const int qrow = qi < len ? qi : len - 1;
    half8 qf[KS];
    {
        const _Float16* qp = p.q + (start + qrow) * p.q_st + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qp + 16 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see prefill.h: keeps the compiler's own vmcnt waits out of the loop)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[bk][r] = 0.f;
    const int q_lo = m * QB + 32 * w;  // first query of this wave
    const half2v ones = {(_Float16)1.f, (_Float16)1.f};

    // S(0): everything issued so far has landed (the wait above); all waves' parts after the barrier
    lds_barrier();
    floatx16 sA[2], sB[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const char* kp = smem + SM::K_OFF + kfrag(ks);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            sA[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(kp + 32 * kb * D * 2), qf[ks], sA[kb], 0, 0, 0);
    }

    // One step: `cur` = S(u) (complete), `nxt` = S(u + 1) (computed here when the half exists).
    // (S(u + 1) of the last step is computed from whatever its slot holds and never used: one code path, no tail variant)
    auto step = [&](int u, floatx16 (&cur)[2], floatx16 (&nxt)[2]) {
        // ---- top of the step: K(u + 1) and V(u) of every wave landed; slot (u + 3) & 3 is free --------------------
        if (u + 2 < nsteps) wait_vm<4>();  // younger: the 2 K + 2 V instructions of half u + 2
        else wait_vm<0>();
        lds_barrier();
        if (u + 3 < nsteps) issue_half(u + 3);
        const int key0 = HALF * u;
        const bool diag = key0 + HALF - 1 > q_lo;  // some key of the half lies beyond some query of the wave
        const char* kst = smem + SM::K_OFF + ((u + 1) & (SM::NST - 1)) * SM::STAGE;
        half8 af[2][2];  // K fragments of two k-steps: those of ks + 1 are read while the MFMAs of ks run
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[kb][r] = 0.f;
            af[0][kb] = *reinterpret_cast<const half8*>(kst + kfrag(0) + 32 * kb * D * 2);
        }
        float mx = -INFINITY, m_new = 0.f, msafe = 0.f, alpha = 1.f, sum = 0.f;
        half8 pb[2][2];
        // ---- phase A: eight slices, each = {K fragments of the next k-step, two QK^T MFMAs of S(u + 1), 1/8 of softmax(S(u))}
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    af[(ks + 1) & 1][kb] = *reinterpret_cast<const half8*>(kst + kfrag(ks + 1) + 32 * kb * D * 2);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                nxt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][kb], qf[ks], nxt[kb], 0, 0, 0);
            if (ks < 2) {  // slices 0, 1: the row maximum of key block ks (causal mask on diagonal halves; S itself is not rewritten)
                const int kb = ks;
                if (diag) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
                        mx = fmaxf(mx, key <= qi ? cur[kb][r] : -INFINITY);  // keys >= len are > every valid query
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, cur[kb][r]);
                }
            } else if (ks == 2) {  // the scaled maximum of the row, the new reference, the rescale factor
                mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2e;
                m_new = fmaxf(m_run, mx);
                msafe = (m_new == -INFINITY) ? 0.f : m_new;
                alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
            } else if (ks < 7) {  // slices 3 .. 6: eight probabilities each (rounded to fp16; row sums over the rounded values)
                const int kb = (ks - 3) >> 1, r0 = 8 * ((ks - 3) & 1);
#pragma unroll
                for (int r = r0; r < r0 + 8; r += 2) {
                    float x0 = cur[kb][r], x1 = cur[kb][r + 1];
                    if (diag) {
                        const int key = key0 + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
                        x0 = key <= qi ? x0 : -INFINITY;
                        x1 = key + 1 <= qi ? x1 : -INFINITY;
                    }
                    const _Float16 p0 = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(x0, p.scale_log2e, -msafe));
                    const _Float16 p1 = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(x1, p.scale_log2e, -msafe));
                    pb[kb][r >> 3][r & 7] = p0;
                    pb[kb][r >> 3][(r & 7) + 1] = p1;
                    const half2v pp = {p0, p1};
                    sum = __builtin_amdgcn_fdot2(pp, ones, sum, false);
                }
            } else {  // slice 7: the row sum
                sum += __shfl_xor(sum, 32);
                l_run = l_run * alpha + sum;
                m_run = m_new;
            }
            __builtin_amdgcn_sched_barrier(0);  // slices stay slices: nothing moves across
        }
        // ---- phase B: O *= alpha, O^T += V(u)^T P(u)^T ---------------------------------------------------------------
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[bk][r] *= alpha;
        }
        const int vst = SM::V_OFF + (u & (SM::NST - 1)) * SM::STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int bk = 0; bk < 4; ++bk) {
                    typedef __attribute__((address_space(3))) short4v* lds_s4;
                    const int vb = vfrag_b[bk] + vst + (32 * kb * D * 2 + (16 * tt) * D * 2);
                    union {
                        short4v s4[2];
                        half8 h8;
                    } av;
                    av.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                    av.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
                    o[bk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av.h8, pb[kb][tt], o[bk], 0, 0, 0);
                }
            }
        }
    };

    for (int u = 0; u < nsteps; u += 2) {
        step(u, sA, sB);
        step(u + 1, sB, sA);
    }
    wait_vm<0>();
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
