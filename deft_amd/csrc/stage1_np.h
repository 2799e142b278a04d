// Stage 1 (Flatten and Node), tile-parallel form: one workgroup per CHUNK of KV tiles, two workgroups per CU,
// placed by the hardware dispatcher.
//
// Included by deft_kernels.hip after plan_records.h, plan_kernels.h and merge.h.
//
// A persistent streaming form (round 1) reached the practical HBM rate in its steady state but
// paid ~8 us per launch around it: a dependent ramp, and a tail in which persistent workgroups that prefetch five
// tiles ahead cannot rebalance.  A plain gather dispatched by the hardware (tools/probes/gather_dma.hip) has neither.
// This kernel keeps that shape and still folds the shared prefix:
//
//   * a workgroup = 4 waves = one chunk: a leader record and its followers (np_record_order), all tiles of ONE run,
//     i.e. with one query list, taken at a stride through the run;
//   * every wave is its own pipeline.  Wave w stages keys [32w, 32w+32) of the tile -- 8 KB of K and 8 KB of V --
//     into its private LDS slices with global_load_lds_dwordx4 and consumes exactly those rows, so there is no
//     barrier between load and use, only its own counted s_waitcnt.  K(i+1) is issued as soon as QK^T(i) is done,
//     V(i+1) after PV(i);
//   * the softmax is wave-private as well: S^T = K Q^T puts the wave's 32 keys on the MFMA M dimension, so a lane
//     holds 16 key scores of one query; P^T (fp16) stays in registers and IS the B operand of O^T += V^T P^T
//     (the contraction order of keys is permuted identically on the V^T side, read with ds_read_b64_tr_b16);
//     each wave keeps (m, l, O[128 x 32]) for its key subset over the whole chunk;
//   * at the end the four waves merge their states through LDS (two barriers per WORKGROUP, not per tile) and write
//     one normalised partial row per virtual query row; the rows of follower tiles are dead in the plan (row_q = -1);
//   * no record staging: a workgroup's record index is its block index, so the descriptor (scalar load), the row
//     offsets and masks (4-byte LDS-DMA into a per-wave area) and nothing else stand between launch and the first
//     K/V request.  Records beyond the leaders exit at once; they sit at the end of the grid.
//
// The merge stays a launch of its own (merge_kernel).  Round 2 built and measured the alternative -- merge workgroups
// inside this launch, waiting on per-KV-head arrival counters for the write-through partial rows -- in four forms
// (profiles/r2_fused_merge_negative.txt): correct, bit-identical, and 3-30 us per layer SLOWER
// than two launches on every workload: a waiting workgroup holds a slot stage 1 wants, a thousand pollers delay
// every arrival, and a cross-workgroup read of fresh rows moves a few GB/s per wave.
#pragma once

namespace deft {

struct NpParams {
    Stage1Params s;
    const char* plan;  // [cap+1][PLAN_BYTES], leaders first (np_record_order)
    const int32_t* hdr;  // plan header (plan_records.h): hdr[1] = number of chunk leaders
    int mirror;        // capped grids: workgroup b takes items b, 2W-1-b, 2W+b, 4W-1-b, ... (the SHORT first items get the extra ones)
    int fast_n;        // workgroups < fast_n (the ones resident at launch) request tile 0's offsets before anything else
#ifdef DEFT_EXPERIMENTS
    int head_rot;      // heads of record r rotated by r * head_rot
    int skew_full;     // > 0: only the first 8 x skew_full workgroups take item = index; behind them the EVEN ones (index % 8 = XCD) take the remaining items, the odd ones none
#endif
    // fused paged append (optional): rows whose plan offset has bit 63 set are read from k_new / v_new
    const _Float16* k_new;
    const _Float16* v_new;
    const int32_t* cache_loc;
    int64_t new_st;
    int n_new;
    unsigned long long* dbg;  // internal: per-workgroup wall-clock stamps [workgroup][8], or null
    // fused rotary embedding (stage1_np_kernel<D, true>; NeoX pairing over the whole head_dim): q rows are rotated as they
    // become MFMA fragments, this step's k rows on their way into the pool and inside the tiles that read them from k_new
    const float* cos_sin;  // [n_new][D] fp32: cos(D/2) | sin(D/2) of new row j = query row j (deft_rope_gather_rows)
};

// o1 = x1 cos - x2 sin, o2 = x2 cos + x1 sin in fp32 without FMA contraction, one rounding to fp16: the arithmetic of
// rope_qk_kernel (deft_kernels.hip) and of oracle/rope.py, so that the fused form is bit-identical to rope + decode.
__device__ __forceinline__ void rope_pair(float x1, float x2, float cs, float sn, _Float16& o1, _Float16& o2) {
#pragma clang fp contract(off)
    const float a = x1 * cs, b = x2 * sn, c = x2 * cs, d = x1 * sn;
    o1 = (_Float16)(a - b);
    o2 = (_Float16)(c + d);
}

// One 16-byte chunk (8 halves at d = 8 ch ..) of a row rotated against its partner chunk ch ^ 8 (d +- D/2):
// cs = cos values of d & (D/2 - 1), sn = the sines.
__device__ __forceinline__ uintx4 rope_chunk(uintx4 own_u, uintx4 par_u, bool first_half, floatx4 c0, floatx4 c1, floatx4 s0,
                                             floatx4 s1) {
    union {
        uintx4 u;
        half8 h8;
    } own, par, res;
    own.u = own_u;
    par.u = par_u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float cs = e < 4 ? c0[e & 3] : c1[e & 3], sn = e < 4 ? s0[e & 3] : s1[e & 3];
        _Float16 o1, o2;
        if (first_half) rope_pair((float)own.h8[e], (float)par.h8[e], cs, sn, o1, o2);
        else rope_pair((float)par.h8[e], (float)own.h8[e], cs, sn, o2, o1);
        res.h8[e] = o1;
    }
    return res.u;
}

// x combined with the value 32 lanes away (lane l with lane l ^ 32) -- gfx950's v_permlane32_swap instead of a ds_bpermute round
// trip through the LDS: the row maximum and the row sum of the wave-private softmax sit on the tile's dependent chain.
__device__ __forceinline__ float max_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int D>
struct NpSmem {
    static constexpr int SLICE = 32 * D * 2;            // one wave's 32 keys of K (or V)
    static constexpr int K_OFF = 0;                     // [4][SLICE]
    static constexpr int V_OFF = 4 * SLICE;             // [4][SLICE]
    static constexpr int Q_OFF = 8 * SLICE;             // Q rows [32][D] fp16, chunks XOR-ed by (row & 15)
    static constexpr int AUX_OFF = Q_OFF + MQ * D * 2;  // per wave 2 slots x 512 B: int64 rowoff[32] | u32 vmask[32] | i32 qsrc[32]
    static constexpr int AUX_SLOT = 512;
    static constexpr int X_OFF = AUX_OFF + 4 * 2 * AUX_SLOT;  // float m[2][4][32], l[2][4][32] (per head of the row: two with head_dim 64)
    static constexpr int OROW_OFF = X_OFF + 2 * 2 * 4 * MQ * 4;   // int32 orow[32] of the leader record
    static constexpr int NEXT_OFF = OROW_OFF + 2 * MQ * 4;    // int32 next work item (orow is staged twice: one 64-lane DMA)
    static constexpr int BYTES = NEXT_OFF + 16;
    static_assert(2 * BYTES <= 160 * 1024, "two workgroups per CU");
};

// Profiling hooks exist in the experiments build only (make exp: -DDEFT_EXPERIMENTS); the shipped kernel has neither
// the ablation branches nor the per-workgroup time stamps.
#ifdef DEFT_EXPERIMENTS
#define ABL(bit) (p.ablate & (bit))
#define DBG np.dbg
#else
#define ABL(bit) false
#define DBG ((unsigned long long*)nullptr)
#endif

// PEEL: Q fragments built in front of the tile loop (what ROPE needs; for the plain kernel measured neutral, tools/ab.py:
// north-star 35.96 / 36.10 us, ToT-50 18.41 / 18.20, Llama-3 north-star tree 17.93 / 17.66 -- the loop form stays)
// NT: K / V rows arrive by non-temporal LDS-DMA (tree modes: a row is read by the few passes of its tile and never again)
// DYN (with NT; GQA launches): the cache policy is the chunk leader's (desc[6]: a tile folded by many
//      passes wants its rows in L2) and a capped grid may take its further items in mirrored order.  Kept out of the plain
//      instantiation: the two wave-uniform branches cost the north-star launch 0.2-0.4 us (tools/ab_rules.sh, late round 4).
// HD2: head_dim 64 (the reference also takes 16 / 32 / 64, tree_attention.py:100, :582).  Two ADJACENT KV heads share one 256-byte
//      pool row -- [slot][K|V][Hkv][64] with heads contiguous -- so a work item is a head PAIR and everything that moves data is
//      the head_dim-128 kernel unchanged: the same DMA granules, the same LDS slices, the same fragment reads.  Only the
//      arithmetic splits: k-steps 0-3 are head A's S^T, 4-7 head B's (two accumulators, two softmaxes), column blocks 0-1 of O^T
//      take head A's probabilities, 2-3 head B's, and the epilogue writes two 64-float partial rows per virtual row.
template <int D, bool ROPE, bool NT, bool PEEL = ROPE, bool HD2 = false, bool DYN = false>
__global__ __launch_bounds__(256, 2) void stage1_np_kernel(NpParams np) {
    constexpr int KS = D / 16;
    constexpr int LPT = 32 * (D / 8) / 64;  // DMA instructions per wave per K (or V) slice
    static_assert(D == 128 && LPT == 8, "tile-parallel stage 1 is instantiated for 256-byte rows (head_dim 128, or two heads of 64)");
    static_assert(!(HD2 && ROPE), "the fused rotary embedding is head_dim 128 only");
    constexpr int NH = HD2 ? 2 : 1;  // heads per row
    using SM = NpSmem<D>;
    const Stage1Params& p = np.s;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    const int bid = blockIdx.x;
    const int W = (int)gridDim.x;
    unsigned long long t_start = 0, t_k0 = 0, t_epi = 0, t_bar = 0;
#ifdef DEFT_EXPERIMENTS
    // per-phase wall-clock sums over the tiles of an item (wave 0; written behind the per-item stamps, tools/np_phases.py):
    // 0 wait K(i) | 1 masks + QK^T + maxima | 2 offsets of tile i+1, K(i+1) / aux(i+2) requests | 3 softmax | 4 wait V(i) | 5 PV, V(i+1) request
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = 0;
#define PHASE(k)                                         \
    if (DBG) {                                           \
        const unsigned long long now_ = wall_clock64(); \
        ph[k] += now_ - ph_t;                            \
        ph_t = now_;                                     \
    }
#define PHASE_START() \
    if (DBG) ph_t = wall_clock64();
#else
#define PHASE(k)
#define PHASE_START()
#endif
    const int HP = HD2 ? p.Hkv / 2 : p.Hkv;                  // rows (heads, or head pairs) per token
    const int64_t kv_shp = HD2 ? 2 * p.kv_sh : p.kv_sh;      // elements between two of them

    // ---- fused paged append: new-token row j is copied into the pool by workgroup (grid-1-j) % grid (nobody reads
    //      those pool rows in this launch: rows flagged NEW in the plan are taken from k_new / v_new) ------------
    for (int copy_job = W - 1 - bid; copy_job < np.n_new; copy_job += W) {
        const int64_t dst = (int64_t)np.cache_loc[copy_job] * p.kv_ss;
        const int chunks = HP * (D / 8);
        const float* cs_row = nullptr;
        if constexpr (ROPE) cs_row = np.cos_sin + (int64_t)copy_job * D;
        for (int i = tid; i < chunks; i += blockDim.x) {
            const int hd = i / (D / 8), ch = i - hd * (D / 8);
            const int64_t so = (int64_t)copy_job * np.new_st + hd * D + ch * 8;
            const int64_t d_o = dst + (int64_t)hd * kv_shp + ch * 8;
            uintx4 kk = *reinterpret_cast<const uintx4*>(np.k_new + so);
            const uintx4 vv = *reinterpret_cast<const uintx4*>(np.v_new + so);
            if constexpr (ROPE) {  // the k row enters the pool rotated: chunk ch pairs with chunk ch ^ 8 (d, d + D/2)
                union {
                    uintx4 u;
                    half8 h8;
                } own, par, res;
                own.u = kk;
                par.u = *reinterpret_cast<const uintx4*>(np.k_new + so + (ch < D / 16 ? D / 2 : -(D / 2)));
                const float* cs = cs_row + 8 * (ch & (D / 16 - 1));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 o1, o2;
                    if (ch < D / 16) rope_pair((float)own.h8[e], (float)par.h8[e], cs[e], cs[D / 2 + e], o1, o2);
                    else rope_pair((float)par.h8[e], (float)own.h8[e], cs[e], cs[D / 2 + e], o2, o1);
                    res.h8[e] = o1;
                }
                kk = res.u;
            }
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.k) + d_o) = kk;
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.v) + d_o) = vv;
        }
    }

    // Work items = (chunk leader record, KV head), head fastest: neighbours share a record and a stretch of the pool,
    // long shared-prefix chunks come first.  Item `bid` is this workgroup's first; a workgroup whose index has
    // further items (capped grids) takes item + W, item + 2 W, ... in a loop.
    int NI = 0x7fffffff;  // leaders x heads, read with the first item's descriptor
    int item = bid, round = 0;
#ifdef DEFT_EXPERIMENTS
    // (DEFT_NP_XCDSKEW: the odd XCDs stream ~18 % slower -- more items to the even ones.  The timeline evens out, the layer gains 2 % at
    //  ONE setting and nothing at its neighbours, profiles/r6_union_len_sweep.txt (4): not shipped)
    bool dead = false;
    if (np.skew_full > 0) {
        const int r = bid >> 3, x = bid & 7;
        if (r >= np.skew_full) {
            if (x & 1) dead = true;
            else item = 8 * np.skew_full + 4 * (r - np.skew_full) + (x >> 1);
        }
    }
#else
    constexpr bool dead = false;
#endif
    auto next_item = [&](int it) __attribute__((always_inline)) {
        if constexpr (DYN) {
            if (np.mirror) return (++round & 1) ? (round + 1) * W - 1 - bid : round * W + bid;  // b, 2W-1-b, 2W+b, 4W-1-b, ...
        }
        return it + W;
    };
    int rec0 = 0, kvh = 0, fb = 0, sd4 = 0, sd0 = 0, sd5 = 0;
    auto rec_of = [&](int i) { return np.plan + (int64_t)(i == 0 ? rec0 : fb + i - 1) * PLAN_BYTES; };

    // ---- loop-invariant lane constants ------------------------------------------------------------------
    const int dpos = l & 15, dkey = l >> 4;
    int kchunk_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk_b[j] = ((dpos ^ dkey) ^ (4 * j)) * 16;
    const int vchunk_b = (dpos ^ (4 * (dkey & 3))) * 16;
    const uint32_t ldsK = SM::K_OFF + (uint32_t)w * SM::SLICE;
    const uint32_t ldsV = SM::V_OFF + (uint32_t)w * SM::SLICE;
    const uint32_t aux0 = SM::AUX_OFF + (uint32_t)w * 2u * SM::AUX_SLOT;
    // S^T A fragments: key row c of this wave's slice, chunk (2ks + h) ^ (c & 15)
    const int krow_b = SM::K_OFF + w * SM::SLICE + c * D * 2;
    const int kcol_b = ((h ^ c) & 15) * 16;
    // O^T A fragments (transpose reads).  MFMA k-index 8h' + e of k-step t is key 16t + 4h' + e (e < 4) and
    // 16t + 8 + 4h' + (e - 4): exactly the keys whose probabilities lane (c, h') holds in acc registers 8t .. 8t+7.
    // Lane (tg = l>>4, tx = l&15) addresses row 4(tg>>1) + (tx>>2) [+16t, +8], columns 32 blk + 16(tg&1) + 4(tx&3),
    // 16-byte chunk XOR-ed by 4*(row & 3) = 4*(tx>>2).
    const int tg = l >> 4, tx = l & 15;
    const int vtr_row_b = SM::V_OFF + w * SM::SLICE + (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    int vtr_col_b[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) vtr_col_b[blk] = (4 * (blk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;

    constexpr int64_t NEW_ROW = (int64_t)1 << 63;
    const char *kb_pool = nullptr, *vb_pool = nullptr, *kb_new = nullptr, *vb_new = nullptr;  // per work item (KV head)

    int64_t rowoff[LPT];
    bool has_new = true;  // (wave-uniform) some row of the slice whose offsets `rowoff` holds is one of this step's new rows
    auto issue_aux = [&](int i, int slot) {  // 2 DMA: this wave's 32 row offsets; its 32 key masks | the 32 q offsets
        const char* rec = rec_of(i);
        dma4(rec + PLAN_ROWOFF + 32 * w * 8 + 4 * l, aux0 + (uint32_t)slot * SM::AUX_SLOT);
        const char* src2 = (l < 32) ? rec + PLAN_MASK + (32 * w + l) * 4 : rec + PLAN_QSRC + (l - 32) * 4;
        dma4(src2, aux0 + (uint32_t)slot * SM::AUX_SLOT + 256u);
    };
    // (round 4: the instruction stream of a tile, not its memory traffic, is what a CU spends most of a small launch on --
    //  profiles/r4_pair_kernel_negative.txt: 1.44 us per tile with no K / V request at all, ~350 VALU instructions per wave at 4
    //  cycles each.  A tile's slice almost never holds one of this step's new rows, so the per-request select between the pool and
    //  k_new / v_new -- compare, two selects, a mask, 16 times per tile -- is decided once per slice: lane k looks at the sign of
    //  row k's offset, one ballot.)
    auto load_rowoff = [&](int slot) {
        const int64_t* ro = reinterpret_cast<const int64_t*>(smem + aux0 + slot * SM::AUX_SLOT);
#pragma unroll
        for (int i = 0; i < LPT; ++i) rowoff[i] = ro[4 * i + dkey];
        const int32_t hi = reinterpret_cast<const int32_t*>(ro)[2 * (l & 31) + 1];
        has_new = ABL(1024) || __builtin_amdgcn_ballot_w64(hi < 0) != 0ull;
    };
    // NT kernels: a chunk whose tiles are folded by many passes (the plan's desc[6]) asks for its rows with the ordinary policy
    bool temporal = false;  // (wave-uniform, per work item)
    auto issue_k_as = [&](auto ntc) __attribute__((always_inline)) {
        constexpr bool nt = decltype(ntc)::value;
        if (!has_new) {
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                if (ABL(8)) continue;  // (experiments: 8 no K / V requests at all; 128 every row is the pool's first -- cache hits)
                const char* src = ABL(128) ? kb_pool : kb_pool + rowoff[i];
                if constexpr (nt) dma16nt(src + kchunk_b[i & 3], ldsK + (uint32_t)i * 1024u);
                else dma16(src + kchunk_b[i & 3], ldsK + (uint32_t)i * 1024u);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            if (ABL(8)) continue;
            const char* src = ABL(128) ? kb_pool : rowoff[i] < 0 ? kb_new + (rowoff[i] & ~NEW_ROW) : kb_pool + rowoff[i];
            if constexpr (nt) dma16nt(src + kchunk_b[i & 3], ldsK + (uint32_t)i * 1024u);
            else dma16(src + kchunk_b[i & 3], ldsK + (uint32_t)i * 1024u);
        }
    };
    auto issue_v_as = [&](auto ntc) __attribute__((always_inline)) {
        constexpr bool nt = decltype(ntc)::value;
        if (!has_new) {
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                if (ABL(8)) continue;
                const char* src = ABL(128) ? vb_pool : vb_pool + rowoff[i];
                if constexpr (nt) dma16nt(src, ldsV + (uint32_t)i * 1024u);
                else dma16(src, ldsV + (uint32_t)i * 1024u);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            if (ABL(8)) continue;
            const char* src = ABL(128) ? vb_pool : rowoff[i] < 0 ? vb_new + (rowoff[i] & ~NEW_ROW) : vb_pool + rowoff[i];
            if constexpr (nt) dma16nt(src, ldsV + (uint32_t)i * 1024u);
            else dma16(src, ldsV + (uint32_t)i * 1024u);
        }
    };
    auto issue_k = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the slice being overwritten
        if (NT && !(DYN && temporal)) issue_k_as(std::true_type{});
        else issue_k_as(std::false_type{});
    };
    auto issue_v = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (NT && !(DYN && temporal)) issue_v_as(std::true_type{});
        else issue_v_as(std::false_type{});
    };
    auto issue_q = [&]() {  // rows 8w .. 8w+7 of the shared Q buffer, offsets from aux slot 0
        const int32_t* qs = reinterpret_cast<const int32_t*>(smem + aux0 + 256 + 128);
        const char* hb = reinterpret_cast<const char*>(p.q) + (int64_t)(HD2 ? 2 * kvh : kvh) * p.G * p.q_sh * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * w + 4 * i + dkey;
            const int chunk = dpos ^ (row & 15);
            // (HD2: the LDS row is [head A's 64 | head B's 64]; B's query head is G heads after A's)
            const int64_t src = HD2 ? (chunk < 8 ? chunk * 16 : (int64_t)p.G * p.q_sh * 2 + (chunk - 8) * 16) : chunk * 16;
            dma16(hb + (int64_t)qs[row] * 2 + src, SM::Q_OFF + (uint32_t)(8 * w + 4 * i) * 256u);
        }
    };

    for (bool first = true;; first = false) {
    if (DBG) t_start = wall_clock64();
    // (wave-uniform by construction, made so explicitly: the quotient comes out of the vector ALU)
    rec0 = __builtin_amdgcn_readfirstlane(item / HP);
    kvh = __builtin_amdgcn_readfirstlane(item - (item / HP) * HP);  // (HD2: the head PAIR)
#ifdef DEFT_EXPERIMENTS
    // (DEFT_NP_HEADROT: the heads of record r handed out rotated by r -- workgroup index % 8 = XCD, so with 32 KV heads every XCD otherwise
    //  reads the same four heads of every record, i.e. the same address bits 8..12 of every row)
    if (np.head_rot) kvh = __builtin_amdgcn_readfirstlane((kvh + rec0 * np.head_rot) % HP);
#endif
    const char* rec_lead = np.plan + (int64_t)rec0 * PLAN_BYTES;
    // The workgroups that are resident when the launch starts set the ramp: for them tile 0's offsets / masks /
    // partial rows are requested (LDS-DMA) before anything is known about the record -- its address only depends
    // on the block index, every record slot of the grid is allocated memory -- so that the leader count, the
    // descriptor and the offsets come back in ONE round trip instead of three dependent ones.  Later workgroups
    // find the plan in L2 and keep the order in which a slot without work exits after a single load.
    const bool spec = first && bid < np.fast_n;  // uniform
    if (spec) {
        if (w == 0) dma4(rec_lead + PLAN_OROW + 4 * (l & 31), SM::OROW_OFF);
        issue_aux(0, 0);
    }
    // The descriptor (and, once, the leader count) by hand-written SCALAR loads, all in flight together, one wait.  Left to the
    // compiler these are vector loads -- the kernel's own stores might alias them -- each followed by s_waitcnt vmcnt(0): two or
    // three dependent round trips in front of every work item (round 3: seen in the ISA; a workgroup's ramp was 1.7-2.2 us).
    typedef int32_t int8v __attribute__((ext_vector_type(8)));
    int8v dsc;
    if (first) {
        int32_t nl;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(dsc), "=&s"(nl)
                     : "s"(rec_lead + PLAN_DESC), "s"(np.hdr + 1)
                     : "memory");
        NI = nl * HP;
        if (item >= NI || dead) {  // this record slot leads no chunk (speculative read of valid memory): nothing to do
            if (spec) wait_vm<0>();
            break;
        }
    } else {
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dsc) : "s"(rec_lead + PLAN_DESC) : "memory");
    }
    sd4 = dsc[4];
    sd0 = dsc[0];
    sd5 = dsc[5];
    if constexpr (DYN) temporal = NT && __builtin_amdgcn_readfirstlane(dsc[6]) != 0;
    const int n = sd4;  // tiles of this chunk (> 0: items only name leaders)
    const int nv = sd0;
    fb = sd5;
    const uint32_t rowbits = nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u);  // the chunk's live virtual rows
    // the scale enters as |scale| inside the exp2 argument (one fma per score instead of a multiply and a subtract); a negative one
    // -- nobody passes it, the C ABI takes any float -- flips the sign of the Q fragments instead
    const float sc = fmaxf(fabsf(p.scale_log2e), 1e-30f);  // (never 0: a masked score is -inf, and -inf * 0 is not)
    kb_pool = reinterpret_cast<const char*>(p.k) + (int64_t)kvh * kv_shp * 2;
    vb_pool = reinterpret_cast<const char*>(p.v) + (int64_t)kvh * kv_shp * 2 + vchunk_b;
    kb_new = reinterpret_cast<const char*>(np.k_new) + (int64_t)kvh * D * 2;
    vb_new = reinterpret_cast<const char*>(np.v_new) + (int64_t)kvh * D * 2 + vchunk_b;
    // leader's partial rows (one per virtual query row), parked in LDS for the epilogue (wave 0, one DMA)
    if (!spec) {
        if (w == 0) dma4(rec_lead + PLAN_OROW + 4 * (l & 31), SM::OROW_OFF);
        issue_aux(0, 0);
    }
    // ---- prologue: aux(0) -> Q, K(0), aux(1), V(0) -----------------------------------------------------
    wait_vm<0>();
    load_rowoff(0);
    // fused rotary embedding of Q: every wave rotates, in LDS, the 8 rows it stages itself -- lane (dpos, dkey) owns
    // position dpos of rows 8 w + dkey and 8 w + 4 + dkey, i.e. source chunk dpos ^ (row & 15), partner chunk at position
    // dpos ^ 8 -- before the barrier that publishes the rows.  The cos | sin values of its two chunks are plain loads
    // issued IN FRONT of the Q / K / V DMA (older, so they have landed when K(0) has; the compiler's own wait at their first
    // use also drains V(0) and aux(1), which makes the counted waits of the first tile conservative).  Measured against
    // inline-asm loads + a counted wait that lets the rotation run while K(0) / V(0) are in flight: that form was 0.3-0.9 us
    // per layer faster on the small trees and 1.6 us SLOWER on the north-star tree (tools/rope_fused_ab.py); not kept.
    floatx4 rq[ROPE ? 8 : 1];
    if constexpr (ROPE) {
        const int32_t* qs = reinterpret_cast<const int32_t*>(smem + aux0 + 256 + 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * w + 4 * i + dkey;
            const int ch = dpos ^ (row & 15);
            const int qrow = qs[row] / (int)p.q_st;
            const float* cs = np.cos_sin + (int64_t)qrow * D + 8 * (ch & 7);
            rq[4 * i + 0] = *reinterpret_cast<const floatx4*>(cs);
            rq[4 * i + 1] = *reinterpret_cast<const floatx4*>(cs + 4);
            rq[4 * i + 2] = *reinterpret_cast<const floatx4*>(cs + D / 2);
            rq[4 * i + 3] = *reinterpret_cast<const floatx4*>(cs + D / 2 + 4);
        }
    }

    issue_q();
    issue_k();
    if (n > 1) issue_aux(1, 1);
    issue_v();

    constexpr bool QF_LDS = HD2;  // Q fragments re-read from LDS every tile instead of living in 32 registers (with ROPE too it measured slower than three spilled registers: gqa_4kx32 fused 23.0 -> 24.3 us per layer)
    half8 qf[QF_LDS ? 1 : KS];
    float m_run[NH], l_run[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) m_run[hh] = -INFINITY, l_run[hh] = 0.f;
    floatx16 o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;

    if constexpr (ROPE || PEEL) {
        // Q rows are rotated and the fragments built in front of the tile loop (the cos / sin registers die here).
        if (n > 1) wait_vm<LPT + 2>();  // K(0) landed, hence this wave's Q rows
        else wait_vm<LPT>();
        if (DBG) t_k0 = wall_clock64();
        if constexpr (ROPE) {
            uintx4 own[2], par[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* rowp = smem + SM::Q_OFF + (8 * w + 4 * i + dkey) * 256;
                own[i] = *reinterpret_cast<const uintx4*>(rowp + dpos * 16);
                par[i] = *reinterpret_cast<const uintx4*>(rowp + (dpos ^ 8) * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane has read before any lane writes
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 8 * w + 4 * i + dkey;
                const int ch = dpos ^ (row & 15);
                *reinterpret_cast<uintx4*>(smem + SM::Q_OFF + row * 256 + dpos * 16) =
                    rope_chunk(own[i], par[i], ch < 8, rq[4 * i], rq[4 * i + 1], rq[4 * i + 2], rq[4 * i + 3]);
            }
        }
        lds_barrier();  // rotated Q rows of all four waves visible
        if constexpr (!QF_LDS) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[ks] = *reinterpret_cast<const half8*>(smem + SM::Q_OFF + c * D * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
            if (p.scale_log2e < 0.f)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[ks] = -qf[ks];
        }
    }
    for (int i = 0; i < n; ++i) {
        const bool has1 = i + 1 < n, has2 = i + 2 < n;
        const int slot = i & 1;
        PHASE_START();
        // ---- K(i) (and Q) landed: younger than it are aux(i+1) [2] and V(i) [8] -------------------------
        if (ABL(2) && i > 0) {  // (experiments: 2 = no K / V waits after the first tile)
        } else if (has1) wait_vm<LPT + 2>();
        else wait_vm<LPT>();
        if (!ROPE && !PEEL && i == 0) {
            if (DBG) t_k0 = wall_clock64();
            lds_barrier();  // Q rows of all four waves visible
            if constexpr (!QF_LDS) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    qf[ks] = *reinterpret_cast<const half8*>(smem + SM::Q_OFF + c * D * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
                if (p.scale_log2e < 0.f)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) qf[ks] = -qf[ks];
            }
        }
        if constexpr (ROPE) {
            // rows of this wave's K slice that came from k_new (this step's tokens) are rotated in LDS before QK^T:
            // lane (dpos, dkey) of DMA piece j owns the 16 bytes at position dpos of row 4 j + dkey, i.e. source chunk
            // dpos ^ (row & 15); its partner chunk (d +- 64) sits at position dpos ^ 8 of the same row.
            bool mine_new = false;
#pragma unroll
            for (int j = 0; j < LPT; ++j) mine_new |= rowoff[j] < 0;
            if (__builtin_amdgcn_ballot_w64(mine_new) != 0ull) {
                // (rare -- the tiles that hold this step's tokens: a real loop over the eight DMA pieces, offsets re-read
                //  from this tile's aux slot, so that no registers of the tile loop are spent on it)
                const int64_t* ro = reinterpret_cast<const int64_t*>(smem + aux0 + slot * SM::AUX_SLOT);
#pragma unroll 1
                for (int j = 0; j < LPT; ++j) {
                    const int64_t ro_j = ro[4 * j + dkey];
                    if (ro_j < 0) {
                        char* rowp = smem + ldsK + j * 1024 + dkey * 256;
                        const uintx4 own = *reinterpret_cast<const uintx4*>(rowp + dpos * 16);
                        const uintx4 par = *reinterpret_cast<const uintx4*>(rowp + (dpos ^ 8) * 16);
                        const int ch = dpos ^ ((4 * (j & 3) + dkey) & 15);
                        const uint32_t new_idx = (uint32_t)(ro_j & ~NEW_ROW) / (uint32_t)(np.new_st * 2);
                        const float* cs =
                            np.cos_sin + (int64_t)new_idx * D + 8 * (ch & 7);
                        const floatx4 c0 = *reinterpret_cast<const floatx4*>(cs), c1 = *reinterpret_cast<const floatx4*>(cs + 4);
                        const floatx4 s0 = *reinterpret_cast<const floatx4*>(cs + D / 2);
                        const floatx4 s1 = *reinterpret_cast<const floatx4*>(cs + D / 2 + 4);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // both reads of every lane before any write
                        *reinterpret_cast<uintx4*>(rowp + dpos * 16) = rope_chunk(own, par, ch < 8, c0, c1, s0, s1);
                    }
                }
            }
        }
        PHASE(0);
        // Does every live row see every one of this wave's 32 keys -- a shared-prefix tile: no pads, no query that is not on the
        // slots' path?  Then no score needs its mask bit tested (three instructions a score): lane k looks at key k's mask word,
        // one ballot.  (wave-uniform; the last tile of a node, a union group's tiles, the leaf tiles take the general path)
        const uint32_t* masks = reinterpret_cast<const uint32_t*>(smem + aux0 + slot * SM::AUX_SLOT + 256);
        const bool full = !ABL(512) && __builtin_amdgcn_ballot_w64((masks[l & 31] & rowbits) != rowbits) == 0ull;
        // ---- S^T for this wave's 32 keys (HD2: k-steps 0-3 = head A, 4-7 = head B) ---------------------------
        floatx16 acc[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[hh][r] = 0.f;
        if (!ABL(1)) {  // experiments build: 1 skip QK^T, 4 skip PV, 16 skip the epilogue, 32 skip the stores
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 a = *reinterpret_cast<const half8*>(smem + krow_b + (kcol_b ^ (32 * ks)));
                constexpr int HALF = KS / 2;
                floatx16& dst = acc[HD2 ? (ks / HALF) : 0];
                if constexpr (QF_LDS) {  // (the variants that are short of registers re-read the Q fragment: 32 VGPRs for 8 LDS reads a tile)
                    half8 qk = *reinterpret_cast<const half8*>(smem + SM::Q_OFF + c * D * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
                    if (p.scale_log2e < 0.f) qk = -qk;
                    dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qk, dst, 0, 0, 0);
                } else {
                    dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], dst, 0, 0, 0);
                }
            }
        }
        if (!full) {  // key masks of this lane's 16 keys (keys 8 g4 + 4 h + j of the wave's 32): masked scores become -inf
            uintx4 m4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) m4[g4] = *reinterpret_cast<const uintx4*>(masks + 8 * g4 + 4 * h);
#pragma unroll
            for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        acc[hh][r] = ((m4[g4][j] >> c) & 1u) ? acc[hh][r] : -INFINITY;
                    }
        }
        // row maxima over the UNSCALED scores (the scale is positive: it commutes with the maximum)
        float mx[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            mx[hh] = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[hh] = fmaxf(mx[hh], acc[hh][r]);
        }
        PHASE(1);
        // ---- the K slice is free: next tile's row offsets -> K(i+1), aux(i+2) ---------------------------
        if (has1) {
            wait_vm<LPT>();  // aux(i+1) landed (younger: V(i))
            load_rowoff(slot ^ 1);
            issue_k();
            if (has2) issue_aux(i + 2, slot);
        }
        PHASE(2);
        // ---- wave-private online softmax (per head of the row) ------------------------------------------------
        half8 pb[NH][2];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const float mxx = max_xor32(mx[hh]) * sc;  // (rounded once, like every scaled score was: the same maximum as before)
            const float m_new = fmaxf(m_run[hh], mxx);
            const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = (m_run[hh] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[hh] - msafe);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const _Float16 ph = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(acc[hh][r], sc, -msafe));
                pb[hh][r >> 3][r & 7] = ph;
                sum += (float)ph;  // row sums over the ROUNDED probabilities: the weights sum to 1 exactly
            }
            sum = sum_xor32(sum);
            l_run[hh] = l_run[hh] * alpha + sum;
            m_run[hh] = m_new;
            if (i > 0 && __builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (HD2 && (b >> 1) != hh) continue;  // (column blocks 0-1 belong to head A, 2-3 to head B)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
                }
            }
        }
        PHASE(3);
        // ---- V(i) landed: younger are K(i+1) [8] and aux(i+2) [2] ---------------------------------------
        if (ABL(2)) {
        } else if (has2) wait_vm<LPT + 2>();
        else if (has1) wait_vm<LPT>();
        else wait_vm<0>();
        PHASE(4);
        // ---- O^T += V^T P^T: four 32-column blocks x two 16-key steps ------------------------------------
        if (!ABL(4))
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                typedef __attribute__((address_space(3))) short4v* lds_s4;
                const int vb = vtr_row_b + vtr_col_b[blk] + (16 * t) * D * 2;
                union {
                    short4v s4[2];
                    half8 h8;
                } av;
                av.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                av.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
                o[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av.h8, pb[HD2 ? (blk >> 1) : 0][t], o[blk], 0, 0, 0);
            }
        }
        if (has1) issue_v();  // V(i+1); rowoff still holds tile i+1's offsets
        PHASE(5);
    }

    if (ABL(16)) {  // (experiments build: no epilogue at all)
        const int next = next_item(item);
        if (next >= NI) break;
        lds_barrier();
        item = next;
        continue;
    }
    if (DBG) t_epi = wall_clock64();
    // ---- epilogue: merge the four waves' (m, l, O) and write one partial row per virtual query row.  ONE barrier:
    //      every wave parks its unscaled O (and m, l) in its own slices -- nobody else ever touched them -- and
    //      the readers rescale while they sum.  Rows of follower tiles are dead by construction: the plan gave
    //      them row_q = -1, so nothing is written for them.
    float* xm = reinterpret_cast<float*>(smem + SM::X_OFF);  // [NH][32 rows][4 waves]: a row's four values are one 16-byte read
    float* xl = xm + 2 * 4 * MQ;
    if (h == 0) {
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            xm[(hh * MQ + c) * 4 + w] = m_run[hh];
            xl[(hh * MQ + c) * 4 + w] = l_run[hh];
        }
    }
    // query row c < 16 in the K slice, c >= 16 in the V slice; [row][128] floats, 16-byte chunk index XOR-ed by the row
    if (c < nv && !(ABL(64))) {
        char* dst = smem + (c < 16 ? SM::K_OFF : SM::V_OFF) + w * SM::SLICE + (c & 15) * (D * 4);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k4 = 8 * blk + 2 * j + h;  // d = 32 blk + 8 j + 4 h + (0..3)
                floatx4 v4 = {o[blk][4 * j], o[blk][4 * j + 1], o[blk][4 * j + 2], o[blk][4 * j + 3]};
                *reinterpret_cast<floatx4*>(dst + ((k4 ^ c) & 31) * 16) = v4;
            }
    }
    if (!(ABL(64))) lds_barrier();
    if (DBG) t_bar = wall_clock64();
    const int32_t* orow = reinterpret_cast<const int32_t*>(smem + SM::OROW_OFF);
    // wave w sums the 16-byte chunks 8w .. 8w+7 of every row: lane = (row within a group of 8, chunk).  HD2: chunks 0-15 are
    // head A's 64 floats (waves 0, 1), chunks 16-31 head B's (waves 2, 3): each with its own head's (m, l), its own output row.
    const int k4 = 8 * w + (l & 7);
    const int hh_out = HD2 ? (w >> 1) : 0;
    const int64_t head_rows = (int64_t)((HD2 ? 2 * kvh + hh_out : kvh)) * p.G * p.rows;
    // The four groups of eight rows are independent, and they have to LOOK independent to the compiler: a loop over the live
    // groups with `if (row < nv)` around each body compiles to four exec-masked blocks executed one after the other, each a
    // chain of six dependent LDS round trips (round 3, seen in the ISA: 1.3 us of a lone wave's 1.75 us epilogue).  So every
    // group's reads are issued unconditionally -- rows >= nv hold stale bytes, all of them valid LDS -- and only the stores are
    // predicated; two groups at a time.  The arithmetic of a live row is unchanged.
#pragma unroll 1
    for (int g0 = 0; 8 * g0 < nv; g0 += 2) {  // two groups in flight (four cost 23-40 spilled registers)
        floatx4 mw[2], lw[2], b[2][4], res[2];
        float lse[2];
        int orow_q[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int qr = 8 * (g0 + g) + (l >> 3);
            const int off = (qr < 16 ? SM::K_OFF : SM::V_OFF) + (qr & 15) * (D * 4) + ((k4 ^ qr) & 31) * 16;
            mw[g] = *reinterpret_cast<const floatx4*>(xm + (hh_out * MQ + qr) * 4);
            lw[g] = *reinterpret_cast<const floatx4*>(xl + (hh_out * MQ + qr) * 4);
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
            // (the values are pinned OUTSIDE the predicated stores: otherwise the compiler sinks the arithmetic and the reads
            //  back into the exec-masked blocks)
            asm volatile("" : "+v"(res[g]), "+v"(lse[g]));
        }
        if (ABL(32)) continue;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int qr = 8 * (g0 + g) + (l >> 3);
            if (qr < nv) {
                const int64_t row = head_rows + orow_q[g];
                // (ordinary stores: non-temporal partial stores, and non-temporal loads of them in the merge, each cost the
                //  north-star layer ~1 us and both ~2.3 -- profiles/r2n_nontemporal.txt)
                if constexpr (HD2) {
                    *reinterpret_cast<floatx4*>(p.partial_o + row * (D / 2) + 4 * (k4 & 15)) = res[g];
                    if ((k4 & 15) == 0) p.partial_lse[row] = lse[g];
                } else {
                    *reinterpret_cast<floatx4*>(p.partial_o + row * D + 4 * k4) = res[g];
                    if (k4 == 0) p.partial_lse[row] = lse[g];
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
        d[4] = (unsigned long long)n;
        d[5] = ((unsigned long long)xcc << 32) | hw;
        d[6] = (unsigned long long)kvh;
        d[7] = t_bar;
#ifdef DEFT_EXPERIMENTS
        unsigned long long* d2 = DBG + (int64_t)(8192 + item) * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) d2[k] = ph[k];
#endif
    }
#ifdef DEFT_EXPERIMENTS
#pragma unroll
    for (int k = 0; k < 6; ++k) ph[k] = 0;
#endif
    // Next item of a capped grid: record capacity beyond the chunk leaders would otherwise be launched as workgroups
    // that only find out that they have nothing to do (tens of thousands for the sequential comparator's entries).
    {
        const int next = next_item(item);
        if (next >= NI) break;
        lds_barrier();  // every wave is done reading the others' slices
        item = next;
    }
    }  // work items
}

#undef ABL
#undef DBG
#undef PHASE
#undef PHASE_START

}  // namespace deft
