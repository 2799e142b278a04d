// Stage 1, 8-wave form: one workgroup per CU takes TWO chunk leaders at a time, one on each half of its eight waves, and when
// the two are the passes 2j, 2j + 1 of one shared node -- the same KV tiles under different 32-row sets of virtual query rows,
// which is what GQA and nodes of more than 32 queries produce (tree_attention.py:894 runs one program per QUERY head; this
// kernel folds the group into a tile's rows, and a 128-key tile under 32 queries x 4 heads has four passes) -- every tile of the
// pair is staged in LDS ONCE and read by both halves.
//
// Included by deft_kernels.hip after stage1_np.h, whose arithmetic this is, fragment for fragment: the same LDS-DMA granules,
// K fragments, transpose reads of V, wave-private softmax, and the same epilogue per half.  Outputs are bit-identical to
// stage1_np_kernel's on the same plan (every partial row is folded from the same keys in the same order by one wave's MFMAs).
//
//   * 512 threads; wave w8 = 4 * half + slice.  Work item `item` = (record pair `item / Hkv`, KV head `item % Hkv`): half 0 takes
//     leader record 2m, half 1 leader record 2m + 1 (plan_kernels.h np_record_order puts the two runs of a pair there; behind the
//     pairs, unpaired leaders sit two by two all the same; a record that leads nothing -- a follower, the slot behind an odd
//     count -- has chunk length 0 and its half idles).  Nothing is looked up: the descriptor of its own record is all a half needs,
//     plus record 2m's pair flag and chunk length, which BOTH halves read, so that the two always agree on the barriers they
//     execute, whatever the plan holds.
//   * two tile buffers of 64 KB (K | V, four 8 KB slices each).
//     PAIRED item: the buffers are a double buffer shared by the halves; every wave requests half of its slice's K rows and half
//     of its V rows -- per CU HALF the L2 -> LDS traffic of two 4-wave workgroups folding the same two passes, and, what
//     matters more (first build: with one tile of lookahead a paired tile took the 2.2 us a 4-wave tile takes, the round trip),
//     the LDS the second copy would have taken is LOOKAHEAD: two barriers per tile -- "K(i) complete" in front of QK^T(i),
//     "V(i) complete" in front of PV(i) -- V(i + 1) is requested behind the first (everyone is past PV(i - 1)), K(i + 2) behind
//     the second (everyone is past QK^T(i)): a tile and a half ahead, up to 96 KB in flight per CU.
//     TWO INDEPENDENT items: half h owns buffer h and runs stage1_np_kernel's own pipeline on it (K(i + 1) requested when QK^T(i)
//     is done, V(i + 1) after PV(i), counted waits, no barrier in the tile loop): two 4-wave workgroups sharing a launch slot.
//   * still two waves per SIMD -- the round-3 forms that shared staged tiles gave that up (one workgroup of four waves per CU:
//     nothing ran under a wave's own MFMA / LDS / exp2 latencies) and lost 35-50 %.
#pragma once

namespace deft {

template <int D>
struct PairSmem {
    static constexpr int SLICE = 32 * D * 2;                   // one wave's 32 keys of K (or V)
    static constexpr int BUF = 8 * SLICE;                      // one tile: K [4][SLICE] | V [4][SLICE]
    static constexpr int KV_OFF = 0;                           // [2][BUF]
    static constexpr int Q_OFF = 2 * BUF;                      // [2 halves][32 rows][D] fp16, chunks XOR-ed by (row & 15)
    static constexpr int Q_HALF = MQ * D * 2;
    static constexpr int AUX_OFF = Q_OFF + 2 * Q_HALF;         // per wave 3 slots x 512 B: int64 rowoff[32] | u32 vmask[32] | i32 qsrc[32]
    static constexpr int AUX_SLOT = 512;
    static constexpr int AUX_SLOTS = 3;                        // (a paired item keeps the offsets / masks of tiles i, i + 1, i + 2)
    static constexpr int X_OFF = AUX_OFF + 8 * AUX_SLOTS * AUX_SLOT;  // per half: float m[32][4], l[32][4]
    static constexpr int X_HALF = 2 * 4 * MQ * 4;
    static constexpr int OROW_OFF = X_OFF + 2 * X_HALF;        // per half: int32 orow[32] of its leader record (staged twice: one 64-lane DMA)
    static constexpr int BYTES = OROW_OFF + 2 * 2 * MQ * 4;
    static_assert(BYTES <= 160 * 1024, "one workgroup per CU");
};

#ifdef DEFT_EXPERIMENTS
#define ABL(bit) (p.ablate & (bit))
#define DBG np.dbg
#else
#define ABL(bit) false
#define DBG ((unsigned long long*)nullptr)
#endif

template <int D, bool NT>
__global__ __launch_bounds__(512, 1) void stage1_pair_kernel(NpParams np) {
    constexpr int KS = D / 16;
    constexpr int LPT = 32 * (D / 8) / 64;  // DMA instructions per wave per K (or V) slice
    static_assert(D == 128 && LPT == 8, "256-byte rows");
    using SM = PairSmem<D>;
    const Stage1Params& p = np.s;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hsel = w8 >> 2;  // half of the workgroup
    const int w = w8 & 3;      // slice of a tile: keys [32 w, 32 w + 32)
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    const int bid = blockIdx.x;
    const int W = (int)gridDim.x;
    unsigned long long t_start = 0, t_k0 = 0, t_epi = 0, t_bar = 0;
    const int HP = p.Hkv;

    // ---- fused paged append (stage1_np_kernel's: new-token row j is copied into the pool by workgroup (grid-1-j) % grid) ----
    for (int copy_job = W - 1 - bid; copy_job < np.n_new; copy_job += W) {
        const int64_t dst = (int64_t)np.cache_loc[copy_job] * p.kv_ss;
        const int chunks = HP * (D / 8);
        for (int i = tid; i < chunks; i += blockDim.x) {
            const int hd = i / (D / 8), ch = i - hd * (D / 8);
            const int64_t so = (int64_t)copy_job * np.new_st + hd * D + ch * 8;
            const int64_t d_o = dst + (int64_t)hd * p.kv_sh + ch * 8;
            const uintx4 kk = *reinterpret_cast<const uintx4*>(np.k_new + so);
            const uintx4 vv = *reinterpret_cast<const uintx4*>(np.v_new + so);
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.k) + d_o) = kk;
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.v) + d_o) = vv;
        }
    }

    int NI = 0x7fffffff;  // record pairs x heads, read with the first item's descriptor
    int item = bid;
    int rec_mine = 0, kvh = 0, fb = 0;
    auto rec_of = [&](int i) { return np.plan + (int64_t)(i == 0 ? rec_mine : fb + i - 1) * PLAN_BYTES; };

    // ---- loop-invariant lane constants (stage1_np_kernel's, relative to a tile buffer) ----------------------------------
    const int dpos = l & 15, dkey = l >> 4;
    int kchunk_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk_b[j] = ((dpos ^ dkey) ^ (4 * j)) * 16;
    const int vchunk_b = (dpos ^ (4 * (dkey & 3))) * 16;
    const uint32_t sliceK = (uint32_t)w * SM::SLICE, sliceV = 4u * SM::SLICE + (uint32_t)w * SM::SLICE;
    const uint32_t aux0 = SM::AUX_OFF + (uint32_t)w8 * (uint32_t)(SM::AUX_SLOTS * SM::AUX_SLOT);
    const uint32_t q_half = SM::Q_OFF + (uint32_t)hsel * SM::Q_HALF;
    const uint32_t orow_half = SM::OROW_OFF + (uint32_t)hsel * (2 * MQ * 4);
    const int krow_b = w * SM::SLICE + c * D * 2;
    const int kcol_b = ((h ^ c) & 15) * 16;
    const int tg = l >> 4, tx = l & 15;
    const int vtr_row_b = 4 * SM::SLICE + w * SM::SLICE + (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    int vtr_col_b[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) vtr_col_b[blk] = (4 * (blk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;

    constexpr int64_t NEW_ROW = (int64_t)1 << 63;
    const char *kb_pool = nullptr, *vb_pool = nullptr, *kb_new = nullptr, *vb_new = nullptr;

    int64_t rowoff[LPT];
    auto issue_aux = [&](int i, int slot) {  // 2 DMA: this wave's 32 row offsets; its 32 key masks | the 32 q offsets
        const char* rec = rec_of(i);
        dma4(rec + PLAN_ROWOFF + 32 * w * 8 + 4 * l, aux0 + (uint32_t)slot * SM::AUX_SLOT);
        const char* src2 = (l < 32) ? rec + PLAN_MASK + (32 * w + l) * 4 : rec + PLAN_QSRC + (l - 32) * 4;
        dma4(src2, aux0 + (uint32_t)slot * SM::AUX_SLOT + 256u);
    };
    auto load_rowoff = [&](int slot) {
        const int64_t* ro = reinterpret_cast<const int64_t*>(smem + aux0 + slot * SM::AUX_SLOT);
#pragma unroll
        for (int i = 0; i < LPT; ++i) rowoff[i] = ro[4 * i + dkey];
    };
    auto issue_k = [&](uint32_t buf) {  // this wave's K slice of a tile -> tile buffer at byte offset `buf`
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of whatever the requests overwrite
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            if (ABL(8)) continue;
            const char* src = ABL(128) ? kb_pool : rowoff[i] < 0 ? kb_new + (rowoff[i] & ~NEW_ROW) : kb_pool + rowoff[i];
            if constexpr (NT) dma16nt(src + kchunk_b[i & 3], buf + sliceK + (uint32_t)i * 1024u);
            else dma16(src + kchunk_b[i & 3], buf + sliceK + (uint32_t)i * 1024u);
        }
    };
    auto issue_v = [&](uint32_t buf) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            if (ABL(8)) continue;
            const char* src = ABL(128) ? vb_pool : rowoff[i] < 0 ? vb_new + (rowoff[i] & ~NEW_ROW) : vb_pool + rowoff[i];
            if constexpr (NT) dma16nt(src, buf + sliceV + (uint32_t)i * 1024u);
            else dma16(src, buf + sliceV + (uint32_t)i * 1024u);
        }
    };
    auto issue_q = [&]() {  // rows 8w .. 8w+7 of this half's Q rows, offsets from aux slot 0
        const int32_t* qs = reinterpret_cast<const int32_t*>(smem + aux0 + 256 + 128);
        const char* hb = reinterpret_cast<const char*>(p.q) + (int64_t)kvh * p.G * p.q_sh * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * w + 4 * i + dkey;
            const int chunk = dpos ^ (row & 15);
            dma16(hb + (int64_t)qs[row] * 2 + chunk * 16, q_half + (uint32_t)(8 * w + 4 * i) * 256u);
        }
    };

    for (bool first = true;; first = false) {
    if (DBG) t_start = wall_clock64();
    const int pair_i = __builtin_amdgcn_readfirstlane(item / HP);
    kvh = __builtin_amdgcn_readfirstlane(item - pair_i * HP);
    rec_mine = 2 * pair_i + hsel;
    const char* rec_lead = np.plan + (int64_t)rec_mine * PLAN_BYTES;
    const char* rec_a = np.plan + (int64_t)(2 * pair_i) * PLAN_BYTES;
    // (stage1_np_kernel: the workgroups resident at launch request tile 0's offsets before anything is known about the record)
    const bool spec = first && bid < np.fast_n;  // uniform
    if (spec) {
        if (w == 0) dma4(rec_lead + PLAN_OROW + 4 * (l & 31), orow_half);
        issue_aux(0, 0);
    }
    // this half's descriptor, record 2m's (the pair flag and chunk length BOTH halves go by), once the leader count: scalar loads,
    // all in flight together, one wait
    typedef int32_t int8v __attribute__((ext_vector_type(8)));
    int8v dsc, dsa;
    if (first) {
        int32_t nl;
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(dsc), "=&s"(dsa), "=&s"(nl)
                     : "s"(rec_lead + PLAN_DESC), "s"(rec_a + PLAN_DESC), "s"(np.hdr + 1)
                     : "memory");
        NI = ((nl + 1) >> 1) * HP;
        if (item >= NI) {  // no leader here (speculative reads of valid memory): nothing to do
            if (spec) wait_vm<0>();
            break;
        }
    } else {
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(dsc), "=&s"(dsa)
                     : "s"(rec_lead + PLAN_DESC), "s"(rec_a + PLAN_DESC)
                     : "memory");
    }
    const bool paired = dsa[6] == 1 && dsa[4] > 0;  // workgroup-uniform: both halves read record 2m
    const int n = paired ? (int)dsa[4] : (int)dsc[4];  // tiles of this half's chunk (0: this record leads nothing -- the half idles)
    const int nv = n > 0 ? (int)dsc[0] : 0;
    fb = dsc[5];
    const bool idle = n <= 0;
    kb_pool = reinterpret_cast<const char*>(p.k) + (int64_t)kvh * p.kv_sh * 2;
    vb_pool = reinterpret_cast<const char*>(p.v) + (int64_t)kvh * p.kv_sh * 2 + vchunk_b;
    kb_new = reinterpret_cast<const char*>(np.k_new) + (int64_t)kvh * D * 2;
    vb_new = reinterpret_cast<const char*>(np.v_new) + (int64_t)kvh * D * 2 + vchunk_b;
    if (!spec && !idle) {
        if (w == 0) dma4(rec_lead + PLAN_OROW + 4 * (l & 31), orow_half);
        issue_aux(0, 0);
    }

    half8 qf[KS];
    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    auto load_qf = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = *reinterpret_cast<const half8*>(smem + q_half + c * D * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
    };

    // One tile folded from the buffer at byte offset `buf`: S^T = K Q^T for this wave's 32 keys, wave-private online softmax,
    // O^T += V^T P^T (stage1_np_kernel's arithmetic).  `after_qk` runs when the K slice has been read, `before_pv` in front of the
    // first read of the V slice.
    // key masks of this lane's 16 keys (keys 8 g4 + 4 h + j of the wave's 32), from a tile's aux slot
    auto load_masks = [&](uintx4 (&m4)[4], int slot) {
        const uint32_t* masks = reinterpret_cast<const uint32_t*>(smem + aux0 + slot * SM::AUX_SLOT + 256);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) m4[g4] = *reinterpret_cast<const uintx4*>(masks + 8 * g4 + 4 * h);
    };
    auto fold_tile = [&](uint32_t buf, const uintx4 (&m4)[4], bool rescale, auto&& after_qk, auto&& before_pv) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!ABL(1)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 a = *reinterpret_cast<const half8*>(smem + buf + krow_b + (kcol_b ^ (32 * ks)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], acc, 0, 0, 0);
            }
        }
        float s[16];
        float mx = -INFINITY;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g4 + j;
                s[r] = ((m4[g4][j] >> c) & 1u) ? acc[r] * p.scale_log2e : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
        after_qk();
        half8 pb[2];
        {
            const float mxx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mxx);
            const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const _Float16 ph = (_Float16)__builtin_amdgcn_exp2f(s[r] - msafe);
                pb[r >> 3][r & 7] = ph;
                sum += (float)ph;  // row sums over the ROUNDED probabilities: the weights sum to 1 exactly
            }
            sum += __shfl_xor(sum, 32);
            l_run = l_run * alpha + sum;
            m_run = m_new;
            if (rescale && __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            }
        }
        before_pv();
        if (!ABL(4))
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                typedef __attribute__((address_space(3))) short4v* lds_s4;
                const int vb = (int)buf + vtr_row_b + vtr_col_b[blk] + (16 * t) * D * 2;
                union {
                    short4v s4[2];
                    half8 h8;
                } av;
                av.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                av.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
                o[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av.h8, pb[t], o[blk], 0, 0, 0);
            }
        }
    };
    auto nothing = []() {};

    if (paired) {
        // ---- both halves fold the SAME tiles from a shared double buffer; every wave requests pieces 4 half .. 4 half + 3 of its
        //      slice (rows 16 half .. 16 half + 15 of the slice's 32), of K and of V ----------------------------------------------
        int64_t ro4[4];  // row offsets of the four pieces this wave requests, of the tile whose K went out last
        auto load_ro4 = [&](int slot) {
            const int64_t* ro = reinterpret_cast<const int64_t*>(smem + aux0 + slot * SM::AUX_SLOT);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ro4[jj] = ro[4 * (4 * hsel + jj) + dkey];
        };
        auto issue_k4 = [&](uint32_t buf) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (ABL(8)) continue;
                const char* src = ABL(128) ? kb_pool : ro4[jj] < 0 ? kb_new + (ro4[jj] & ~NEW_ROW) : kb_pool + ro4[jj];
                if constexpr (NT) dma16nt(src + kchunk_b[jj], buf + sliceK + (uint32_t)(4 * hsel + jj) * 1024u);
                else dma16(src + kchunk_b[jj], buf + sliceK + (uint32_t)(4 * hsel + jj) * 1024u);
            }
        };
        auto issue_v4 = [&](uint32_t buf) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (ABL(8)) continue;
                const char* src = ABL(128) ? vb_pool : ro4[jj] < 0 ? vb_new + (ro4[jj] & ~NEW_ROW) : vb_pool + ro4[jj];
                if constexpr (NT) dma16nt(src, buf + sliceV + (uint32_t)(4 * hsel + jj) * 1024u);
                else dma16(src, buf + sliceV + (uint32_t)(4 * hsel + jj) * 1024u);
            }
        };
        // Program order of this wave's requests: aux(0) aux(1) aux(2) | Q K(0) V(0) K(1) | then per tile i: V(i+1) aux(i+3) K(i+2)
        // (each only if that tile exists).  The counted waits below follow from it.
        if (n > 1) issue_aux(1, 1);
        if (n > 2) issue_aux(2, 2);
        wait_vm<0>();
        load_ro4(0);
        issue_q();
        issue_k4(0u);
        issue_v4(0u);
        if (n > 1) {
            load_ro4(1);
            issue_k4((uint32_t)SM::BUF);
        }
        int s3 = 0;  // aux slot of tile i (i % 3)
        for (int i = 0; i < n; ++i) {
            const bool has1 = i + 1 < n, has2 = i + 2 < n, has3 = i + 3 < n;
            const uint32_t buf = (uint32_t)(i & 1) * SM::BUF;
            // ---- K(i) (and, i = 0, the Q rows) landed.  Younger: i = 0: V(0) K(1); else V(i) aux(i+2) K(i+1) ---------------
            if (i == 0) {
                if (has1) wait_vm<8>();
                else wait_vm<4>();
            } else if (has2) wait_vm<10>();
            else if (has1) wait_vm<8>();
            else wait_vm<4>();
            if (DBG && i == 0) t_k0 = wall_clock64();
            lds_barrier();  // K(i) complete -- and every wave past PV(i - 1): the V part of the other buffer is free
            if (has1) issue_v4(buf ^ (uint32_t)SM::BUF);  // V(i+1); ro4 holds tile i + 1's offsets
            if (i == 0) load_qf();
            uintx4 m4[4];
            load_masks(m4, s3);
            if (has3) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the masks of tile i have been read
                issue_aux(i + 3, s3);
            }
            fold_tile(
                buf, m4, i > 0, nothing,
                [&]() {
                    // ---- V(i) landed, and the offsets of tile i + 2.  Younger than V(i): i = 0: K(1) V(1) aux(3); else
                    //      aux(i+2) K(i+1) V(i+1) aux(i+3) -- with a tile i + 2, aux(i+2) has to be in as well --------------
                    if (has2) {  // (may stay in flight: K(i+1) V(i+1) aux(i+3))
                        if (has3) wait_vm<10>();
                        else wait_vm<8>();
                    } else if (has1) wait_vm<8>();
                    else wait_vm<0>();
                    lds_barrier();  // V(i) complete -- and every wave past QK^T(i): the K part of this buffer is free
                    if (has2) {
                        const int s2 = s3 == 0 ? 2 : s3 - 1;  // (i + 2) % 3
                        load_ro4(s2);
                        issue_k4(buf);  // K(i+2)
                    }
                });
            s3 = s3 == 2 ? 0 : s3 + 1;
        }
        lds_barrier();  // every wave is done with the last tile: the buffers become the epilogue's parking space
    } else {
        // ---- two independent items: half h owns buffer h; stage1_np_kernel's pipeline ---------------------------------------
        const uint32_t buf = (uint32_t)hsel * SM::BUF;
        if (!idle) {
            // prologue: aux(0) -> Q, K(0), aux(1), V(0)
            wait_vm<0>();
            load_rowoff(0);
            issue_q();
            issue_k(buf);
            if (n > 1) issue_aux(1, 1);
            issue_v(buf);
            if (n > 1) wait_vm<LPT + 2>();  // K(0) landed, hence this wave's Q rows
            else wait_vm<LPT>();
        } else if (spec) {
            wait_vm<0>();
        }
        if (DBG) t_k0 = wall_clock64();
        lds_barrier();  // Q rows of a half's four waves visible (both halves: the barrier is the workgroup's)
        if (!idle) {
            load_qf();
            for (int i = 0; i < n; ++i) {
                const bool has1 = i + 1 < n, has2 = i + 2 < n;
                const int slot = i & 1;
                // K(i) landed: younger than it are aux(i+1) [2] and V(i) [8]
                if (i > 0) {
                    if (ABL(2)) {
                    } else if (has1) wait_vm<LPT + 2>();
                    else wait_vm<LPT>();
                }
                uintx4 m4[4];
                load_masks(m4, slot);
                fold_tile(
                    buf, m4, i > 0,
                    [&]() {  // the K slice is free: next tile's row offsets -> K(i+1), aux(i+2)
                        if (has1) {
                            wait_vm<LPT>();  // aux(i+1) landed (younger: V(i))
                            load_rowoff(slot ^ 1);
                            issue_k(buf);
                            if (has2) issue_aux(i + 2, slot);
                        }
                    },
                    [&]() {  // V(i) landed: younger are K(i+1) [8] and aux(i+2) [2]
                        if (ABL(2)) {
                        } else if (has2) wait_vm<LPT + 2>();
                        else if (has1) wait_vm<LPT>();
                        else wait_vm<0>();
                    });
                if (has1) issue_v(buf);  // V(i+1); rowoff still holds tile i+1's offsets
            }
        }
    }

    if (DBG) t_epi = wall_clock64();
    // ---- epilogue, per half (stage1_np_kernel's): the half's four waves park their unscaled O (and m, l) in the half's own
    //      buffer, one barrier, then each wave rescales-and-sums a quarter of the columns of every row --------------------------
    float* xm = reinterpret_cast<float*>(smem + SM::X_OFF + hsel * SM::X_HALF);  // [32 rows][4 waves]
    float* xl = xm + 4 * MQ;
    const uint32_t park = (uint32_t)hsel * SM::BUF;
    if (!idle) {
        if (h == 0) {
            xm[c * 4 + w] = m_run;
            xl[c * 4 + w] = l_run;
        }
        // query row c < 16 in the K part, c >= 16 in the V part; [row][128] floats, 16-byte chunk index XOR-ed by the row
        if (c < nv) {
            char* dst = smem + park + (c < 16 ? 0 : 4 * SM::SLICE) + w * SM::SLICE + (c & 15) * (D * 4);
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k4 = 8 * blk + 2 * j + h;  // d = 32 blk + 8 j + 4 h + (0..3)
                    floatx4 v4 = {o[blk][4 * j], o[blk][4 * j + 1], o[blk][4 * j + 2], o[blk][4 * j + 3]};
                    *reinterpret_cast<floatx4*>(dst + ((k4 ^ c) & 31) * 16) = v4;
                }
        }
    }
    lds_barrier();
    if (DBG) t_bar = wall_clock64();
    if (!idle) {
        const int32_t* orow = reinterpret_cast<const int32_t*>(smem + orow_half);
        const int k4 = 8 * w + (l & 7);
        const int64_t head_rows = (int64_t)kvh * p.G * p.rows;
        // (two independent groups of eight rows in flight, reads unconditional, stores predicated: see stage1_np_kernel)
#pragma unroll 1
        for (int g0 = 0; 8 * g0 < nv; g0 += 2) {
            floatx4 mw[2], lw[2], b[2][4], res[2];
            float lse[2];
            int orow_q[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int qr = 8 * (g0 + g) + (l >> 3);
                const int off = (int)park + (qr < 16 ? 0 : 4 * SM::SLICE) + (qr & 15) * (D * 4) + ((k4 ^ qr) & 31) * 16;
                mw[g] = *reinterpret_cast<const floatx4*>(xm + qr * 4);
                lw[g] = *reinterpret_cast<const floatx4*>(xl + qr * 4);
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) b[g][ww] = *reinterpret_cast<const floatx4*>(smem + off + ww * SM::SLICE);
                orow_q[g] = orow[qr];
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float M = fmaxf(fmaxf(mw[g][0], mw[g][1]), fmaxf(mw[g][2], mw[g][3]));
                float L = 0.f;
                floatx4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float f = (mw[g][ww] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw[g][ww] - M);
                    L += f * lw[g][ww];
                    a += b[g][ww] * f;
                }
                const float inv = L > 0.f ? 1.f / L : 0.f;
                res[g] = a * inv;
                lse[g] = (L > 0.f) ? (M + __builtin_amdgcn_logf(L)) * LN2 : -INFINITY;
                asm volatile("" : "+v"(res[g]), "+v"(lse[g]));
            }
            if (ABL(32)) continue;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int qr = 8 * (g0 + g) + (l >> 3);
                if (qr < nv) {
                    const int64_t row = head_rows + orow_q[g];
                    *reinterpret_cast<floatx4*>(p.partial_o + row * D + 4 * k4) = res[g];
                    if (k4 == 0) p.partial_lse[row] = lse[g];
                }
            }
        }
    }
    if (DBG && (tid & 255) == 0 && item < 4096) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = DBG + ((int64_t)item * 2 + hsel) * 8;
        d[0] = t_start;
        d[1] = t_k0;
        d[2] = t_epi;
        d[3] = idle ? 0ull : wall_clock64();
        d[4] = (unsigned long long)(n + (paired ? 100 : 0));
        d[5] = ((unsigned long long)xcc << 32) | hw;
        d[6] = (unsigned long long)kvh;
        d[7] = t_bar;
    }
    if (item + W >= NI) break;
    lds_barrier();  // every wave is done reading the parked rows
    item += W;
    }  // work items
}

#undef ABL
#undef DBG

}  // namespace deft
