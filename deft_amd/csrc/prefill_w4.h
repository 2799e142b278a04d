// Causal prefill attention, 4-wave form: 128 queries per workgroup, TWO workgroups per CU, 64-key stages.
//
// Experiment (experiments build only, DEFT_PREFILL_W4=1).  The shipped kernel (prefill.h) runs 8 waves behind one barrier per
// tile, so the two waves of a SIMD are in the same phase (QK^T, softmax, PV) at the same time and one pipe idles while the
// other works.  Here a CU holds two INDEPENDENT workgroups of 4 waves (one wave of each per SIMD), each with its own barriers,
// free to drift apart -- the arrangement the decode kernel (stage1_np.h) uses.  Price: every K / V stage serves 128 queries
// instead of 256 (twice the L2 -> LDS traffic).  Same arithmetic, LDS row formats and grid order as prefill.h.
#pragma once

namespace deft {

template <int D>
struct PrefillW4Smem {
    static constexpr int HALF = 64;             // keys per stage
    static constexpr int STAGE = HALF * D * 2;  // one K (or V) stage: 16 KB
    static constexpr int K_OFF = 0;             // two stages
    static constexpr int V_OFF = 2 * STAGE;     // two stages
    static constexpr int BYTES = 4 * STAGE;     // 64 KB: two workgroups per CU
};

template <int D>
__global__ __launch_bounds__(256, 2) void prefill_w4_kernel(PrefillParams p) {
    constexpr int KS = D / 16;
    constexpr int QB = 128;  // queries per workgroup
    constexpr int HALF = 64;
    static_assert(D == 128, "prefill is instantiated for head_dim 128");
    using SM = PrefillW4Smem<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    // grid order of prefill.h with 128-query blocks (p.nblk counts THOSE here)
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

    const int dpos = l & 15, dkey = l >> 4;
    const int tg = l >> 4, tx = l & 15;
    const int vtr_row_b = (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    int vfrag_b[4];
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) vfrag_b[bk] = vtr_row_b + (4 * (bk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
    const int krow_b = c * D * 2;
    const int kcol_b = ((h ^ c) & 15) * 16;
    int kfrag_b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfrag_b[ks] = krow_b + (kcol_b ^ (32 * ks));

    const char* kbase = reinterpret_cast<const char*>(p.k + start * p.k_st + (int64_t)kvh * p.k_sh);
    const char* vbase = reinterpret_cast<const char*>(p.v + start * p.v_st + (int64_t)kvh * p.v_sh);
    const uint32_t kS = (uint32_t)(p.k_st * 2), vS = (uint32_t)(p.v_st * 2);
    auto issue_half = [&](int x, int stg) {  // 4 K + 4 V instructions per wave: keys 16 w + 4 i + dkey of half x
        const char* kh = kbase + (int64_t)HALF * x * p.k_st * 2;
        const char* vh = vbase + (int64_t)HALF * x * p.v_st * 2;
        const int last = len - 1 - HALF * x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = 16 * w + 4 * i + dkey;
            const uint32_t kc = (uint32_t)(key < last ? key : last);
            dma16s(kh, kc * kS + (uint32_t)((dpos ^ (key & 15)) * 16), SM::K_OFF + (uint32_t)stg * SM::STAGE + (uint32_t)(16 * w + 4 * i) * 256u);
            dma16s(vh, kc * vS + (uint32_t)((dpos ^ (4 * (key & 3))) * 16), SM::V_OFF + (uint32_t)stg * SM::STAGE + (uint32_t)(16 * w + 4 * i) * 256u);
        }
    };

    const int nsteps = min(2 * m + 2, (len + HALF - 1) / HALF);
    issue_half(0, 0);

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
    const half2v ones = {(_Float16)1.f, (_Float16)1.f};

    for (int u = 0; u < nsteps; ++u) {
        const int stg = u & 1;
        wait_vm<0>();
        lds_barrier();
        if (u + 1 < nsteps) issue_half(u + 1, stg ^ 1);
        const int key0 = HALF * u;
        if (key0 > q_lo + 31) continue;  // the whole stage lies above this wave's queries (wave-uniform)
        const bool diag = key0 + HALF - 1 > q_lo;
        floatx16 acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
        const int kstage = SM::K_OFF + stg * SM::STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const char* kp = smem + (kfrag_b[ks] + kstage);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(kp + 32 * kb * D * 2), qf[ks], acc[kb], 0, 0, 0);
        }
        float mx = -INFINITY;
        if (diag) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
                    acc[kb][r] = key <= qi ? acc[kb][r] : -INFINITY;
                    mx = fmaxf(mx, acc[kb][r]);
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[kb][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2e;
        const float m_new = fmaxf(m_run, mx);
        const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
        half8 pb[2][2];
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const _Float16 p0 = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(acc[kb][r], p.scale_log2e, -msafe));
                const _Float16 p1 = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(acc[kb][r + 1], p.scale_log2e, -msafe));
                pb[kb][r >> 3][r & 7] = p0;
                pb[kb][r >> 3][(r & 7) + 1] = p1;
                const half2v pp = {p0, p1};
                sum = __builtin_amdgcn_fdot2(pp, ones, sum, false);
            }
        sum += __shfl_xor(sum, 32);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[bk][r] *= alpha;
        }
        const int vstage = SM::V_OFF + stg * SM::STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int bk = 0; bk < 4; ++bk) {
                    typedef __attribute__((address_space(3))) short4v* lds_s4;
                    const int vb = vfrag_b[bk] + vstage + (32 * kb * D * 2 + (16 * tt) * D * 2);
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
    }
    wait_vm<0>();
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
