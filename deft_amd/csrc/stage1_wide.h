// Stage 1, MULTI-PASS form: one workgroup per CU stages every KV tile of its chunk ONCE and folds it for up to WIDE_PASSES
// 32-row passes of virtual query rows.
//
// Included by deft_kernels.hip after stage1_np.h (whose lane geometry, MFMA fragments and arithmetic it repeats).
//
// Why.  stage1_np_kernel gives every 32-row pass of a tile its own workgroup.  A tile under more than 32 virtual rows -- GQA
// with many branches (42 leaves x 4 query heads = six passes over a ToT root), a node with more than 32 queries (the reference
// emits it once per 32-query chunk: Medusa-64) -- is then fetched L2 -> LDS once per pass, and a small launch becomes many
// short workgroups that each pay a ramp (two dependent round trips before the first MFMA), an epilogue and, beyond the
// resident slots, a dispatch turn: ToT-50 ran 608 workgroups of 1-4 tiles in 16 us for 32 MB (profiles/r2a_timeline_tot50_4k.txt).
// Here the passes of one tile ("sibling" runs of the plan, plan_kernels.h record_order_parallel) share the staged tile:
//
//   * 4 waves, ONE workgroup per CU (512 registers per lane, the CU's whole LDS): wave w keeps keys [32w, 32w+32) of every
//     tile, as in stage1_np_kernel, but for up to WIDE_PASSES passes -- (m, l, O[128 x 32]) per pass in registers, the K / V^T
//     fragment of a k-step read from LDS once and fed to one MFMA per pass;
//   * K and V slices are DOUBLE-buffered per wave: tile i + 2 is requested as soon as tile i has been consumed, so a wave
//     has two tiles of K and V in flight and never waits for a round trip it could have started earlier (the single-buffered
//     kernel exposes one memory latency per tile: 1.6-2.0 us per tile on the small trees);
//   * no barrier in the tile loop: per-wave LDS-DMA with counted waits.  The DMA stream of a wave is
//         T1 (one trip, with the descriptor): R(0) M(0)* QS*            row offsets, pass-0 masks, pass-0 q offsets
//         T2: QS M(0) OROW R(1) | K(0) V(0) Q0                          sibling passes' masks / offsets (their records are named
//                                                                       by the descriptor), the first tile, pass-0 Q rows
//         T3: R(2) Qs K(1) M(1)    [Q fragments built]    V(1)          sibling Q rows, the second tile
//         iteration i:  ... QK^T, softmax ... R(i+3) K(i+2) ... PV ... V(i+2) M(i+2)
//     with R = 1, M = 2, QS = 2, OROW = 2, K = V = 8 instructions; R and M are issued for tiles beyond the chunk too (they
//     alias the leader's record) so that every wait count is a constant of (has a next tile);
//   * Q rows of all passes are staged through the second V buffers (free until V(1) is requested), two barriers per work item;
//   * epilogue: the waves park their per-pass O in their own K / V buffers two passes at a time and merge across waves as
//     stage1_np_kernel does; partial rows, log-sum-exps and the merge kernel are unchanged.
//
// Results are bit-identical to stage1_np_kernel run on the same plan: the same keys meet the same queries in the same
// order inside a wave, and the cross-wave merge is the same code.
#pragma once

namespace deft {

template <int D>
struct WideSmem {
    static constexpr int PW = WIDE_PASSES;
    static constexpr int SLICE = 32 * D * 2;              // one wave's 32 keys of K (or V): 8 KB
    static constexpr int K_OFF = 0;                       // [4 waves][2 buffers][SLICE]
    static constexpr int V_OFF = 8 * SLICE;               // [4 waves][2 buffers][SLICE]; buffer 1 of wave s stages pass s's Q rows first
    static constexpr int AUX_OFF = 16 * SLICE;            // per wave: R[2][256] | M[2][512] | QS[512] | OROW[512]
    static constexpr int R_SLOT = 256, M_SLOT = 512;
    static constexpr int AUX_R = 0, AUX_M = 2 * R_SLOT, AUX_QS = AUX_M + 2 * M_SLOT, AUX_OROW = AUX_QS + 512;
    static constexpr int AUX_WAVE = AUX_OROW + 512;       // 2.5 KB
    static constexpr int X_OFF = AUX_OFF + 4 * AUX_WAVE;  // float m[2][4][32], l[2][4][32] of the two passes being merged
    static constexpr int BYTES = X_OFF + 2 * 2 * 4 * MQ * 4;
    static_assert(PW <= 4, "Q rows of pass s are staged in wave s's second V buffer");
    static_assert(BYTES <= 160 * 1024, "one workgroup per CU");
};

#ifdef DEFT_EXPERIMENTS
#define DBG np.dbg
#else
#define DBG ((unsigned long long*)nullptr)
#endif

template <int D, bool NT>
__global__ __launch_bounds__(256, 1) void stage1_wide_kernel(NpParams np) {
    constexpr int KS = D / 16;
    constexpr int LPT = 32 * (D / 8) / 64;  // DMA instructions per wave per K (or V) slice
    static_assert(D == 128 && LPT == 8, "instantiated for head_dim 128");
    using SM = WideSmem<D>;
    constexpr int PW = SM::PW;
    const Stage1Params& p = np.s;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    const int bid = blockIdx.x;
    const int W = (int)gridDim.x;
    unsigned long long t_start = 0, t_k0 = 0, t_epi = 0;

    // ---- fused paged append (as stage1_np_kernel): row j is copied into the pool by workgroup (grid-1-j) % grid ----------
    for (int copy_job = W - 1 - bid; copy_job < np.n_new; copy_job += W) {
        const int64_t dst = (int64_t)np.cache_loc[copy_job] * p.kv_ss;
        const int chunks = p.Hkv * (D / 8);
        for (int i = tid; i < chunks; i += blockDim.x) {
            const int hd = i / (D / 8), ch = i - hd * (D / 8);
            const int64_t so = (int64_t)copy_job * np.new_st + hd * D + ch * 8;
            const int64_t d_o = dst + (int64_t)hd * p.kv_sh + ch * 8;
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.k) + d_o) = *reinterpret_cast<const uintx4*>(np.k_new + so);
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.v) + d_o) = *reinterpret_cast<const uintx4*>(np.v_new + so);
        }
    }

    // ---- loop-invariant lane constants (stage1_np_kernel's, with a buffer index added where a slice is addressed) ----------
    const int dpos = l & 15, dkey = l >> 4;
    int kchunk_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk_b[j] = ((dpos ^ dkey) ^ (4 * j)) * 16;
    const int vchunk_b = (dpos ^ (4 * (dkey & 3))) * 16;
    const uint32_t ldsK = SM::K_OFF + (uint32_t)w * 2u * SM::SLICE;
    const uint32_t ldsV = SM::V_OFF + (uint32_t)w * 2u * SM::SLICE;
    const uint32_t aux = SM::AUX_OFF + (uint32_t)w * SM::AUX_WAVE;
    const int krow_b = SM::K_OFF + w * 2 * SM::SLICE + c * D * 2;
    const int kcol_b = ((h ^ c) & 15) * 16;
    const int tg = l >> 4, tx = l & 15;
    const int vtr_row_b = SM::V_OFF + w * 2 * SM::SLICE + (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    int vtr_col_b[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) vtr_col_b[blk] = (4 * (blk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
    constexpr int64_t NEW_ROW = (int64_t)1 << 63;

    int NI = 0x7fffffff;
    for (int item = bid;; item += W) {
        if (DBG) t_start = wall_clock64();
        // (made wave-uniform explicitly: the quotient comes out of the vector ALU, and descriptor reads through a vector address
        //  would be vector loads -- four dependent round trips, counted in vmcnt, instead of scalar loads beside the DMA)
        const int rec0 = __builtin_amdgcn_readfirstlane(item / p.Hkv), kvh = __builtin_amdgcn_readfirstlane(item - (item / p.Hkv) * p.Hkv);
        const char* rec_lead = np.plan + (int64_t)rec0 * PLAN_BYTES;
        // ---- T1: everything that only depends on the item's index, in ONE round trip with the descriptor.  A workgroup's first
        //      item is read before the item count is known: every record slot of the grid is allocated memory. ----------------
        dma4(rec_lead + PLAN_ROWOFF + 32 * w * 8 + 4 * l, aux + SM::AUX_R);
        dma4(rec_lead + PLAN_MASK + (32 * w + c) * 4, aux + SM::AUX_M);        // (both halves: pass 0 -- replaced in T2)
        dma4(rec_lead + PLAN_MASK + (32 * w + c) * 4, aux + SM::AUX_M + 256);
        dma4(rec_lead + PLAN_QSRC + c * 4, aux + SM::AUX_QS);
        dma4(rec_lead + PLAN_QSRC + c * 4, aux + SM::AUX_QS + 256);
        // (scalar loads by hand: the compiler reads descriptors the kernel's own stores might alias with VECTOR loads -- dependent
        //  round trips counted in vmcnt.  Load and wait sit in one asm statement, the DMA above is already in flight.)
        typedef int32_t int8v __attribute__((ext_vector_type(8)));
        int8v d0;
        if (NI == 0x7fffffff) {
            int32_t np_items;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(d0), "=&s"(np_items)
                         : "s"(rec_lead + PLAN_DESC), "s"(np.hdr + HDR_PRIMARIES)
                         : "memory");
            NI = np_items * p.Hkv;
        } else {
            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0) : "s"(rec_lead + PLAN_DESC) : "memory");
        }
        const int n = d0[4];      // tiles of the chunk
        const int fb0 = d0[5];    // its first follower record
        const int sib1 = d0[6];   // leader of the first sibling pass
        const int pnS = d0[7];
        if (item >= NI) {
            wait_vm<0>();
            break;
        }
        const int npass = min(max(pnS & 15, 1), PW);
        const int S = pnS >> 4;
        // leader / first follower / virtual rows of every pass (absent passes alias pass 0 and are never folded)
        int lead[PW], fol[PW], nvp[PW];
        lead[0] = rec0;
        fol[0] = fb0;
        nvp[0] = d0[0];
#pragma unroll
        for (int s = 1; s < PW; ++s) {
            lead[s] = s < npass ? sib1 + (s - 1) * S : rec0;
            fol[s] = fb0;  // (read below, once T2 is on its way)
            nvp[s] = 0;
        }
        auto rec_of = [&](int s, int i) { return np.plan + (int64_t)(i == 0 ? lead[s] : fol[s] + i - 1) * PLAN_BYTES; };
        // record of tile i as seen by this LANE in a two-pass DMA: lanes 0-31 pass 2g, lanes 32-63 pass 2g + 1 (tiles beyond
        // the chunk alias the leaders: their masks are fetched and never read)
        auto lane_rec = [&](int g, int i) {
            const int ii = i < n ? i : 0;
            const char* a = rec_of(2 * g < PW ? 2 * g : 0, ii);
            const char* b = rec_of(2 * g + 1 < PW ? 2 * g + 1 : 0, ii);
            return h ? b : a;
        };
        auto issue_r = [&](int i) {  // this wave's 32 row offsets of tile i
            const char* rec = rec_of(0, i < n ? i : 0);
            dma4(rec + PLAN_ROWOFF + 32 * w * 8 + 4 * l, aux + SM::AUX_R + (uint32_t)(i & 1) * SM::R_SLOT);
        };
        auto issue_m = [&](int i) {  // its 32 key masks for every pass
            const uint32_t dst = aux + SM::AUX_M + (uint32_t)(i & 1) * SM::M_SLOT;
            dma4(lane_rec(0, i) + PLAN_MASK + (32 * w + c) * 4, dst);
            dma4(lane_rec(1, i) + PLAN_MASK + (32 * w + c) * 4, dst + 256);
        };
        const char* kb_pool = reinterpret_cast<const char*>(p.k) + (int64_t)kvh * p.kv_sh * 2;
        const char* vb_pool = reinterpret_cast<const char*>(p.v) + (int64_t)kvh * p.kv_sh * 2 + vchunk_b;
        const char* kb_new = reinterpret_cast<const char*>(np.k_new) + (int64_t)kvh * D * 2;
        const char* vb_new = reinterpret_cast<const char*>(np.v_new) + (int64_t)kvh * D * 2 + vchunk_b;
        int64_t rowoff[LPT];
        auto load_rowoff = [&](int i) {
            const int64_t* ro = reinterpret_cast<const int64_t*>(smem + aux + SM::AUX_R + (i & 1) * SM::R_SLOT);
#pragma unroll
            for (int j = 0; j < LPT; ++j) rowoff[j] = ro[4 * j + dkey];
        };
        auto issue_k = [&](int i) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the buffer being overwritten
            const uint32_t dst = ldsK + (uint32_t)(i & 1) * SM::SLICE;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const char* src = rowoff[j] < 0 ? kb_new + (rowoff[j] & ~NEW_ROW) : kb_pool + rowoff[j];
                if constexpr (NT) dma16nt(src + kchunk_b[j & 3], dst + (uint32_t)j * 1024u);
                else dma16(src + kchunk_b[j & 3], dst + (uint32_t)j * 1024u);
            }
        };
        auto issue_v = [&](int i) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t dst = ldsV + (uint32_t)(i & 1) * SM::SLICE;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const char* src = rowoff[j] < 0 ? vb_new + (rowoff[j] & ~NEW_ROW) : vb_pool + rowoff[j];
                if constexpr (NT) dma16nt(src, dst + (uint32_t)j * 1024u);
                else dma16(src, dst + (uint32_t)j * 1024u);
            }
        };
        // rows 8w .. 8w+7 of pass s's Q buffer (the second V buffer of wave s), chunks XOR-ed by (row & 15)
        const char* qhead = reinterpret_cast<const char*>(p.q) + (int64_t)kvh * p.G * p.q_sh * 2;
        auto issue_q = [&](int s) {
            const int32_t* qs = reinterpret_cast<const int32_t*>(smem + aux + SM::AUX_QS) + 32 * s;
            const uint32_t qbuf = SM::V_OFF + (uint32_t)(2 * s + 1) * SM::SLICE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 8 * w + 4 * i + dkey;
                const int chunk = dpos ^ (row & 15);
                dma16(qhead + (int64_t)qs[row] * 2 + chunk * 16, qbuf + (uint32_t)(8 * w + 4 * i) * 256u);
            }
        };

        wait_vm<0>();  // T1 landed
        load_rowoff(0);
        const bool has1_0 = n > 1;
        // ---- T2 ---------------------------------------------------------------------------------------------------------
        {
            const uint32_t qsd = aux + SM::AUX_QS, ord = aux + SM::AUX_OROW;
            dma4(lane_rec(0, 0) + PLAN_QSRC + c * 4, qsd);
            dma4(lane_rec(1, 0) + PLAN_QSRC + c * 4, qsd + 256);
            issue_m(0);
            dma4(lane_rec(0, 0) + PLAN_OROW + c * 4, ord);
            dma4(lane_rec(1, 0) + PLAN_OROW + c * 4, ord + 256);
            issue_r(1);
        }
        issue_k(0);
        issue_v(0);
        issue_q(0);  // (pass 0's q offsets came with T1)
#pragma unroll
        for (int s = 1; s < PW; ++s) {  // the sibling passes' descriptors, while T2 is in flight
            int8v ds;
            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(ds)
                         : "s"(np.plan + (int64_t)lead[s] * PLAN_BYTES + PLAN_DESC)
                         : "memory");
            fol[s] = ds[5];
            nvp[s] = s < npass ? ds[0] : 0;
        }
        wait_vm<2 * LPT + 2>();  // the small ones landed: sibling q offsets, row offsets of tile 1
        // ---- T3 ---------------------------------------------------------------------------------------------------------
        issue_r(2);
#pragma unroll
        for (int s = 1; s < PW; ++s) issue_q(s);  // (absent passes re-stage pass 0's rows: constant instruction counts)
        if (has1_0) {
            load_rowoff(1);
            issue_k(1);
        }
        issue_m(1);

        half8 qf[PW][KS];
        float m_run[PW], l_run[PW];
        floatx16 o[PW][4];
#pragma unroll
        for (int s = 0; s < PW; ++s) {
            m_run[s] = -INFINITY;
            l_run[s] = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[s][b][r] = 0.f;
        }

        for (int i = 0; i < n; ++i) {
            const bool has1 = i + 1 < n, has2 = i + 2 < n;
            const int buf = i & 1;
            if (i == 0) {
                // K(0), V(0) and every pass's Q rows landed (younger: K(1), M(1)); fragments; then the second V buffers are free
                if (has1) wait_vm<LPT + 2>();
                else wait_vm<2>();
                if (DBG) t_k0 = wall_clock64();
                lds_barrier();
#pragma unroll
                for (int s = 0; s < PW; ++s) {
                    const char* qb = smem + SM::V_OFF + (2 * s + 1) * SM::SLICE;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        qf[s][ks] = *reinterpret_cast<const half8*>(qb + c * D * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
                }
                lds_barrier();
                if (has1) {
                    load_rowoff(1);  // (re-read rather than held: the registers are wanted by the fragments meanwhile)
                    issue_v(1);
                }
            } else {
                // K(i), V(i), M(i) landed: younger are R(i+2), K(i+1), V(i+1), M(i+1)
                if (has1) wait_vm<2 * LPT + 3>();
                else wait_vm<3>();
            }
            // ---- pass after pass over the staged slices (the passes are kept apart on purpose: interleaving them is what the
            //      compiler does unasked, and three passes' fragments in flight at once do not fit the register file) -------------
#pragma unroll
            for (int s = 0; s < PW; ++s) {
                if (s >= npass) continue;
                // S^T for this wave's 32 keys
                floatx16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 a = *reinterpret_cast<const half8*>(smem + krow_b + buf * SM::SLICE + (kcol_b ^ (32 * ks)));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s][ks], acc, 0, 0, 0);
                }
                if (s == npass - 1) {
                    // the K buffer is free: row offsets of tile i + 2 (landed: younger are K(i+1), V(i+1), M(i+1)) -> R(i+3), K(i+2)
                    if (has1) wait_vm<2 * LPT + 2>();
                    else wait_vm<2>();
                    if (has2) load_rowoff(i + 2);
                    issue_r(i + 3);
                    if (has2) issue_k(i + 2);
                }
                // wave-private online softmax (stage1_np_kernel's arithmetic)
                uintx4 m4[4];
                const uint32_t* masks = reinterpret_cast<const uint32_t*>(smem + aux + SM::AUX_M + buf * SM::M_SLOT) + 32 * s;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) m4[g4] = *reinterpret_cast<const uintx4*>(masks + 8 * g4 + 4 * h);
                float sc[16];
                float mx = -INFINITY;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        sc[r] = ((m4[g4][j] >> c) & 1u) ? acc[r] * p.scale_log2e : -INFINITY;
                        mx = fmaxf(mx, sc[r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run[s], mx);
                const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = (m_run[s] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[s] - msafe);
                half8 pb[2];
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const _Float16 ph = (_Float16)__builtin_amdgcn_exp2f(sc[r] - msafe);
                    pb[r >> 3][r & 7] = ph;
                    sum += (float)ph;  // row sums over the ROUNDED probabilities
                }
                sum += __shfl_xor(sum, 32);
                l_run[s] = l_run[s] * alpha + sum;
                m_run[s] = m_new;
                if (i > 0 && __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[s][b][r] *= alpha;
                }
                // O^T += V^T P^T: four 32-column blocks x two 16-key steps
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk) {
                        typedef __attribute__((address_space(3))) short4v* lds_s4;
                        const int vb = vtr_row_b + buf * SM::SLICE + vtr_col_b[blk] + (16 * t) * D * 2;
                        union {
                            short4v s4[2];
                            half8 h8;
                        } av;
                        av.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                        av.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
                        o[s][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av.h8, pb[t], o[s][blk], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has2) {
                load_rowoff(i + 2);  // (re-read: not held across the passes)
                issue_v(i + 2);
            }
            issue_m(i + 2);
        }
        wait_vm<0>();  // (the row offsets / masks requested beyond the chunk)

        if (DBG) t_epi = wall_clock64();
        // ---- epilogue: two passes at a time, every wave parks its unscaled O of pass s in ITS buffers (s & 1), one barrier, the
        //      waves rescale-and-sum a quarter of the columns each (stage1_np_kernel's epilogue) ----------------------------------
        float* xm = reinterpret_cast<float*>(smem + SM::X_OFF);  // [2][4][32]
        float* xl = xm + 2 * 4 * MQ;
        const int64_t head_rows = (int64_t)kvh * p.G * p.rows;
        const int32_t* orow_all = reinterpret_cast<const int32_t*>(smem + aux + SM::AUX_OROW);
        const int k4 = 8 * w + (l & 7);
#pragma unroll
        for (int rd = 0; rd < (PW + 1) / 2; ++rd) {
            if (2 * rd >= npass) break;
            if (rd > 0) lds_barrier();  // the readers of the previous round are done with the buffers
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * rd + e;
                if (s >= PW || s >= npass) continue;
                if (h == 0) {
                    xm[(e * 4 + w) * MQ + c] = m_run[s];
                    xl[(e * 4 + w) * MQ + c] = l_run[s];
                }
                if (c < nvp[s]) {
                    char* dst = smem + (c < 16 ? SM::K_OFF : SM::V_OFF) + (2 * w + e) * SM::SLICE + (c & 15) * (D * 4);
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int kk = 8 * blk + 2 * j + h;  // d = 32 blk + 8 j + 4 h + (0..3)
                            floatx4 v4 = {o[s][blk][4 * j], o[s][blk][4 * j + 1], o[s][blk][4 * j + 2], o[s][blk][4 * j + 3]};
                            *reinterpret_cast<floatx4*>(dst + ((kk ^ c) & 31) * 16) = v4;
                        }
                }
            }
            lds_barrier();
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * rd + e;
                if (s >= PW || s >= npass) continue;
                const int nv = nvp[s];
                for (int q0 = 0; q0 < nv; q0 += 8) {
                    const int qr = q0 + (l >> 3);
                    if (qr < nv) {
                        float mw[4];
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) mw[ww] = xm[(e * 4 + ww) * MQ + qr];
                        const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
                        float L = 0.f;
                        floatx4 a = {0.f, 0.f, 0.f, 0.f};
                        const int off = (qr < 16 ? SM::K_OFF : SM::V_OFF) + e * SM::SLICE + (qr & 15) * (D * 4) + ((k4 ^ qr) & 31) * 16;
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            const float f = (mw[ww] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[ww] - M);
                            L += f * xl[(e * 4 + ww) * MQ + qr];
                            const floatx4 b = *reinterpret_cast<const floatx4*>(smem + off + ww * 2 * SM::SLICE);
                            a += b * f;
                        }
                        const float inv = L > 0.f ? 1.f / L : 0.f;
                        const floatx4 res = a * inv;
                        const float lse = (L > 0.f) ? (M + __builtin_amdgcn_logf(L)) * LN2 : -INFINITY;
                        const int64_t row = head_rows + orow_all[32 * s + qr];
                        *reinterpret_cast<floatx4*>(p.partial_o + row * D + 4 * k4) = res;
                        if (k4 == 0) p.partial_lse[row] = lse;
                    }
                }
            }
        }
        if (DBG && tid == 0 && item < 8192) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* d = DBG + (int64_t)item * 8;
            d[0] = t_start;
            d[1] = t_k0;
            d[2] = t_epi;
            d[3] = wall_clock64();
            d[4] = (unsigned long long)(n + 100 * npass);
            d[5] = ((unsigned long long)xcc << 32) | hw;
            d[6] = (unsigned long long)kvh;
        }
        if (item + W >= NI) break;
        lds_barrier();  // every wave is done with the others' buffers before the next item's DMA
    }
}

#undef DBG

}  // namespace deft
