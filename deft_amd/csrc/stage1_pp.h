// Stage 1 (Flatten, GQA), pass-parallel form: one workgroup per chunk of KV tiles AND group of up to four 32-row passes,
// a WAVE PER PASS.
//
// Included by deft_kernels.hip after stage1_np.h (NpParams, the DMA / wait helpers, the LDS layouts of the K / V rows).
//
// With G query heads per KV head a tile's virtual query rows (query x head of the group) come in passes of 32: the north-star
// tree on Llama-3-8B has 32 x 4 = 128 rows = four passes over every tile of the shared prefix, ToT-50 has seven.  The
// tile-parallel kernel (stage1_np.h) runs every (chunk, pass) as its own workgroup: the same K / V rows travel L2 -> LDS once
// per pass, and four times as many workgroups each pay the ramp, the latency-paced tile chain and the cross-wave epilogue
// (round 3, DESIGN.md section 4f: a lone wave per SIMD spends 1.5 us per tile of which 0.3 are arithmetic).  Here:
//
//   * a work item = (pass group, KV head): the chunk leader of the group's first pass (the plan puts those leaders first,
//     plan_kernels.h np_record_order `pp`) -- its descriptor says where the other passes' records are;
//   * the four waves stage each 128-key tile ONCE, 32 keys each, by LDS-DMA into a double-buffered stage (tile i + 1 lands
//     while tile i is consumed; one barrier per tile) -- the prefill kernel's pipeline (prefill.h);
//   * wave w owns pass w: S^T = K Q^T for ALL 128 keys of the tile (32 MFMAs on four accumulators), one online-softmax
//     step per tile, O^T += V^T P^T (32 MFMAs).  A wave's rows are its own from the first tile to the last, so there is NO
//     cross-wave merge at the end: the epilogue is a normalisation in registers and 16 stores per lane;
//   * waves without a pass (a group of fewer than four) only stage.
//
// Arithmetic as in stage1_np.h (fp16 operands, fp32 accumulation, scale folded into the logits before the maximum, P rounded
// to fp16 with the row sums over the ROUNDED values), but a row's 128 keys of a tile are ONE softmax step here and four
// wave-private ones merged at the end there: the two kernels agree to rounding, not bit for bit.  A launch uses one or the
// other as a function of its geometry alone (launch_stage1_np), so the eager path and a captured session agree exactly.
#pragma once

namespace deft {

struct PpSmem {
    static constexpr int STAGE = TILE * 256;           // one K (or V) tile: 128 rows of 256 bytes
    static constexpr int K_OFF = 0;                    // [2][STAGE]
    static constexpr int V_OFF = 2 * STAGE;            // [2][STAGE]
    static constexpr int AUX_OFF = 4 * STAGE;          // per wave 3 slots x 1 KB
    static constexpr int AUX_SLOT = 1024;              // int64 rowoff[32] (staging rows) | u32 vmask[128] | i32 qsrc[32] | i32 orow[32]
    static constexpr int AUX_MASK = 256, AUX_QSRC = 768, AUX_OROW = 896;
    static constexpr int NSLOT = 3;
    static constexpr int BYTES = AUX_OFF + 4 * NSLOT * AUX_SLOT;  // 140 KB: one workgroup per CU
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

template <bool NT>
__global__ __launch_bounds__(256, 1) void stage1_pp_kernel(NpParams np) {
    constexpr int D = 128, KS = D / 16, LPT = 8;
    using SM = PpSmem;
    const Stage1Params& p = np.s;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, c = l & 31, h = l >> 5;
    const int bid = blockIdx.x, W = (int)gridDim.x;
    const int HP = p.Hkv;

    // ---- fused paged append (stage1_np.h): new-token row j is copied into the pool by workgroup (grid-1-j) % grid ----
    for (int copy_job = W - 1 - bid; copy_job < np.n_new; copy_job += W) {
        const int64_t dst = (int64_t)np.cache_loc[copy_job] * p.kv_ss;
        const int chunks = HP * (D / 8);
        for (int i = tid; i < chunks; i += blockDim.x) {
            const int hd = i / (D / 8), ch = i - hd * (D / 8);
            const int64_t so = (int64_t)copy_job * np.new_st + hd * D + ch * 8;
            const int64_t d_o = dst + (int64_t)hd * p.kv_sh + ch * 8;
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.k) + d_o) = *reinterpret_cast<const uintx4*>(np.k_new + so);
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.v) + d_o) = *reinterpret_cast<const uintx4*>(np.v_new + so);
        }
    }

    // ---- lane constants: the LDS layouts of stage1_np.h / prefill.h (K chunks XOR-ed by key & 15, V chunks by 4*(key & 3)) ----
    const int dpos = l & 15, dkey = l >> 4;
    const int tg = l >> 4, tx = l & 15;
    int kchunk_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk_b[j] = ((dpos ^ dkey) ^ (4 * j)) * 16;
    const int vchunk_b = (dpos ^ (4 * (dkey & 3))) * 16;
    int kfrag_b[KS], vfrag_b[4];
    {
        const int krow_b = c * 256, kcol_b = ((h ^ c) & 15) * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kfrag_b[ks] = krow_b + (kcol_b ^ (32 * ks));
        const int vtr_row_b = (4 * (tg >> 1) + (tx >> 2)) * 256 + (tx & 1) * 8;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) vfrag_b[blk] = vtr_row_b + (4 * (blk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
    }
    const uint32_t aux0 = SM::AUX_OFF + (uint32_t)w * (uint32_t)(SM::NSLOT * SM::AUX_SLOT);
    const uint32_t ldsK = SM::K_OFF + (uint32_t)w * 32u * 256u, ldsV = SM::V_OFF + (uint32_t)w * 32u * 256u;  // this wave's 32 keys of a stage
    constexpr int64_t NEW_ROW = (int64_t)1 << 63;

    int NI = 0x7fffffff;
    int item = bid;
    for (bool first = true;; first = false) {
        const int rec0 = __builtin_amdgcn_readfirstlane(item / HP);
        const int kvh = __builtin_amdgcn_readfirstlane(item - (item / HP) * HP);
        const char* rec_lead = np.plan + (int64_t)rec0 * PLAN_BYTES;
        // the group leader's descriptor (and, once, the item count) by scalar loads, one wait (stage1_np.h)
        typedef int32_t int8v __attribute__((ext_vector_type(8)));
        int8v dsc;
        if (first) {
            int32_t ng, nl;
            asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dword %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(dsc), "=&s"(ng), "=&s"(nl)
                         : "s"(rec_lead + PLAN_DESC), "s"(np.hdr + HDR_GROUPS), "s"(np.hdr + 1)
                         : "memory");
            NI = (ng > 0 ? ng : nl) * HP;  // (a plan without pass groups: every chunk leader is a group of one pass)
            if (item >= NI) break;
        } else {
            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dsc) : "s"(rec_lead + PLAN_DESC) : "memory");
        }
        const int n = dsc[4];                                    // tiles of the chunk, the same in every pass
        const int Pg = (dsc[7] & 7) ? (dsc[7] & 7) : 1;          // passes of the group = waves that compute
        const int nv_last = dsc[7] ? ((dsc[7] >> 3) & 63) : dsc[0];
        const int S = dsc[7] >> 9, F = dsc[2] >> 1;              // leader / follower records between two passes of the run
        const int fb0 = dsc[5];
        const bool active = w < Pg;                              // (uniform per wave)
        const int nv = !active ? 0 : (w == Pg - 1 ? (Pg == 1 ? dsc[0] : nv_last) : MQ);
        // records of tile i: the group leader's chain for the staging rows, this wave's own pass for masks / q rows / partial rows
        const int lead_w = active ? (w == 0 ? rec0 : dsc[6] + (w - 1) * S) : rec0;
        const int fb_w = active ? fb0 + w * F : fb0;
        auto rec_g = [&](int i) { return np.plan + (int64_t)(i == 0 ? rec0 : fb0 + i - 1) * PLAN_BYTES; };
        auto rec_w = [&](int i) { return np.plan + (int64_t)(i == 0 ? lead_w : fb_w + i - 1) * PLAN_BYTES; };

        const char* kb_pool = reinterpret_cast<const char*>(p.k) + (int64_t)kvh * p.kv_sh * 2;
        const char* vb_pool = reinterpret_cast<const char*>(p.v) + (int64_t)kvh * p.kv_sh * 2 + vchunk_b;
        const char* kb_new = reinterpret_cast<const char*>(np.k_new) + (int64_t)kvh * D * 2;
        const char* vb_new = reinterpret_cast<const char*>(np.v_new) + (int64_t)kvh * D * 2 + vchunk_b;

        int64_t rowoff[LPT];
        auto issue_aux = [&](int i, int slot) {  // 3 DMA (4 for tile 0): staging rows, this pass's masks, (q offsets | partial rows)
            const uint32_t dst = aux0 + (uint32_t)slot * SM::AUX_SLOT;
            dma4(rec_g(i) + PLAN_ROWOFF + 32 * w * 8 + 4 * l, dst);
            const char* rw = rec_w(i);
            dma4(rw + PLAN_MASK + 4 * l, dst + SM::AUX_MASK);
            dma4(rw + PLAN_MASK + 256 + 4 * l, dst + SM::AUX_MASK + 256u);
            if (i == 0) dma4(rw + (l < 32 ? PLAN_QSRC + 4 * l : PLAN_OROW + 4 * (l - 32)), dst + SM::AUX_QSRC);
        };
        auto load_rowoff = [&](int slot) {
            const int64_t* ro = reinterpret_cast<const int64_t*>(smem + aux0 + slot * SM::AUX_SLOT);
#pragma unroll
            for (int i = 0; i < LPT; ++i) rowoff[i] = ro[4 * i + dkey];
        };
        auto issue_kv = [&](int stg) {  // 8 K + 8 V instructions: keys 32w + 4i + dkey of the tile whose offsets are in rowoff
            const uint32_t so = (uint32_t)stg * SM::STAGE;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const char* ks = rowoff[i] < 0 ? kb_new + (rowoff[i] & ~NEW_ROW) : kb_pool + rowoff[i];
                const char* vs = rowoff[i] < 0 ? vb_new + (rowoff[i] & ~NEW_ROW) : vb_pool + rowoff[i];
                if constexpr (NT) {
                    dma16nt(ks + kchunk_b[i & 3], ldsK + so + (uint32_t)i * 1024u);
                    dma16nt(vs, ldsV + so + (uint32_t)i * 1024u);
                } else {
                    dma16(ks + kchunk_b[i & 3], ldsK + so + (uint32_t)i * 1024u);
                    dma16(vs, ldsV + so + (uint32_t)i * 1024u);
                }
            }
        };

        // ---- prologue: aux(0), aux(1) -> K(0), V(0), Q fragments ----------------------------------------------------
        issue_aux(0, 0);
        if (n > 1) {
            issue_aux(1, 1);
            wait_vm<3>();
        } else {
            wait_vm<0>();
        }
        load_rowoff(0);
        issue_kv(0);
        half8 qf[KS];
        {
            const int qs = reinterpret_cast<const int32_t*>(smem + aux0 + SM::AUX_QSRC)[c];
            const _Float16* qp = p.q + (int64_t)kvh * p.G * p.q_sh + (active ? qs : 0) + 8 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qp + 16 * ks);
        }
        // (the compiler's own wait for the Q loads, placed here: prefill.h)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

        float m_run = -INFINITY, l_run = 0.f;
        floatx16 o[4];
#pragma unroll
        for (int bk = 0; bk < 4; ++bk)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[bk][r] = 0.f;

        for (int i = 0; i < n; ++i) {
            const int stg = i & 1;
            const int slot = i % SM::NSLOT;
            wait_vm<0>();   // tile i (this wave's rows) and aux(i+1) landed
            lds_barrier();  // ... everyone's rows; and every wave is done with tile i-1's stage
            if (i + 1 < n) {
                load_rowoff((i + 1) % SM::NSLOT);
                issue_kv(stg ^ 1);
                if (i + 2 < n) issue_aux(i + 2, (i + 2) % SM::NSLOT);
            }
            if (!active) continue;
            // ---- S^T for all 128 keys: four accumulator chains, K fragments double-buffered by k-step (prefill.h) --------
            floatx16 acc[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
            const int kstage = SM::K_OFF + stg * SM::STAGE;
            half8 af[2][4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) af[0][kb] = *reinterpret_cast<const half8*>(smem + (kfrag_b[0] + kstage) + 32 * kb * 256);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                    const char* kp = smem + (kfrag_b[ks + 1] + kstage);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) af[(ks + 1) & 1][kb] = *reinterpret_cast<const half8*>(kp + 32 * kb * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][kb], qf[ks], acc[kb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- masks (bit c of the key's word: this lane's row sees the slot), scale, one online-softmax step ------------
            const uint32_t* masks = reinterpret_cast<const uint32_t*>(smem + aux0 + slot * SM::AUX_SLOT + SM::AUX_MASK);
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uintx4 m4 = *reinterpret_cast<const uintx4*>(masks + 32 * kb + 8 * g4 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        acc[kb][r] = ((m4[j] >> c) & 1u) ? acc[kb][r] * p.scale_log2e : -INFINITY;
                        mx = fmaxf(mx, acc[kb][r]);
                    }
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
            half8 pb[4][2];
            float sum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const _Float16 ph = (_Float16)__builtin_amdgcn_exp2f(acc[kb][r] - msafe);
                    pb[kb][r >> 3][r & 7] = ph;
                    sum += (float)ph;  // row sums over the ROUNDED probabilities: the weights sum to 1 exactly
                }
            sum += __shfl_xor(sum, 32);
            l_run = l_run * alpha + sum;
            m_run = m_new;
            if (i > 0 && __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
                for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[bk][r] *= alpha;
            }
            // ---- O^T += V^T P^T: 32 MFMAs, V^T fragments double-buffered by (key block, k-step) group (prefill.h) ------------
            const int vstage = SM::V_OFF + stg * SM::STAGE;
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
                    const int vb = vfrag_b[bk] + vstage + (32 * kb * 256 + (16 * tt) * 256);
                    dst[bk].s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                    dst[bk].s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * 256));
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
        }

        // ---- epilogue: this wave's rows are complete -- normalise in registers, one partial row per live virtual row.
        //      Lane (c, h) holds d = 32 bk + 8 j + 4 h + (0..3) of row c.
        if (active && c < nv) {
            const int orow = reinterpret_cast<const int32_t*>(smem + aux0 + SM::AUX_OROW)[c];
            const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
            const int64_t row = (int64_t)kvh * p.G * p.rows + orow;
            float* po = p.partial_o + row * D + 4 * h;
#pragma unroll
            for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const floatx4 v4 = {o[bk][4 * j] * inv, o[bk][4 * j + 1] * inv, o[bk][4 * j + 2] * inv, o[bk][4 * j + 3] * inv};
                    *reinterpret_cast<floatx4*>(po + 32 * bk + 8 * j) = v4;
                }
            if (h == 0) p.partial_lse[row] = (l_run > 0.f) ? (m_run + __builtin_amdgcn_logf(l_run)) * LN2 : -INFINITY;
        }
        if (item + W >= NI) break;
        lds_barrier();  // every wave is done with the stages and its aux slots
        item += W;
    }
}

}  // namespace deft
