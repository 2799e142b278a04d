// Flatten stage 1, streaming form: persistent workgroups, LDS-DMA pipeline.
//
// Included by deft_kernels.hip (needs its typedefs and Stage1Params).
//
// Why a second form of stage 1: the tile-per-workgroup kernel (stage1_kernel) pays
// three dependent HBM round trips per tile (descriptor -> slot list -> K/V rows)
// with only two workgroups per CU to hide them, spends ~1400 instructions per wave
// per tile, and writes one fp32 partial per (tile, head).  Here
//   * a plan kernel (once per call, or once per decode step when the caller caches
//     the plan) packs each block's metadata into one 2 KB record: the byte offset
//     of every KV row in the pool, a 32-bit query mask per slot (0 for padding) and
//     {cnt, prow, run_start, len};
//   * a fixed grid of workgroups (2 per CU) each walks a contiguous span of the
//     (KV head, tile) sequence, head-major.  A workgroup is 4 compute waves + 4
//     LOADER waves: only the loaders issue DMA, so the compute waves never stall on
//     a full memory queue (measured: with compute waves issuing their own DMA, half
//     of every tile's time was spent blocked in the issue of 16 instructions), and a
//     1 KB DMA instruction costs ~90 cycles to issue, so a tile's 64 are spread over
//     four waves;
//   * K and V tiles travel HBM -> LDS by `global_load_lds_dwordx4` (no VGPR
//     round trip).  K(i+1) is issued as soon as the S^T MFMAs of tile i are done and
//     V(i+1) as soon as its PV MFMAs are done, so a tile's softmax/PV time hides the
//     next K and its QK^T/softmax time hides the next V.  The 2 KB plan record of
//     tile i+2 rides the same DMA queue;
//   * V^T MFMA fragments come from `ds_read_b64_tr_b16` (2 reads per fragment
//     instead of 8 16-bit reads + 4 permutes); V is stored with its 16-byte chunks
//     XOR-ed by 4*(key&3) so the four rows a transpose-read touches sit in different
//     bank groups, K with chunks XOR-ed by key&15 for the row-per-lane b128 reads;
//   * consecutive tiles of one head whose query list is identical (the whole shared
//     prefix of a few-shot tree) are folded with an online softmax in registers
//     and emit ONE partial, which cuts the fp32 partial traffic of the reference
//     (17-19 MB per layer-step on the 4k x 32 tree) to a few MB.
//
// All DMA issue and all waits on it are inline asm: hipcc neither counts asm VMEM
// operations nor drains them at a raw s_barrier, which is what lets loads stay in
// flight across barriers (cdna_hip_programming.md §5.7, "Pipelining across barriers").
// Wait arithmetic (per loader wave, LPT = its DMA instructions per K or V tile = 32*D/8/64):
//   issue order   ... K(i) | meta(i+1) V(i) | K(i+1) | meta(i+2) V(i+1) | ...
//   "K(i) and meta(i+1) landed"  <=>  at most V(i)    outstanding  -> vmcnt(LPT)
//   "V(i) landed"                <=>  at most K(i+1)  outstanding  -> vmcnt(LPT)
//   last tile of the span: nothing younger was issued               -> vmcnt(0)
// Stores are never counted (they may retire early; over-waiting is safe).
#pragma once

namespace deft {

typedef int32_t intx4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

// One plan record per work unit of ONE KV head (+ a sentinel record after the last one).  A unit is a
// 128-slot KV tile together with up to 32 "virtual query rows": row v = (query qi, head g of the GQA
// group), g fastest.  A tile whose cnt * G rows exceed 32 appears once per 32-row pass; passes of a run
// of tiles with one query list are ordered pass-major so that consecutive records fold.
constexpr int PLAN_BYTES = 2048;
constexpr int PLAN_ROWOFF = 0;   // int64[128]  byte offset of each slot's row in the pool (pads alias slot 0)
constexpr int PLAN_MASK = 1024;  // uint32[128] bit v set <=> virtual row v sees the slot (0 for pads)
constexpr int PLAN_DESC = 1536;  // int32[4]    n_vrows, -, run_start, len
constexpr int PLAN_QSRC = 1600;  // int32[32]   element offset of row v's Q vector from q + kvh*G*q_stride_head
constexpr int PLAN_OROW = 1728;  // int32[32]   partial row of row v, relative to kvh*G*rows: g*rows + prow + qi
constexpr int PLAN_HDR = 256;    // plan header: int32 {records per head R, ...}, scheduler words at +64

struct StreamParams {
    Stage1Params s;
    const int32_t* hdr;  // plan header: hdr[0] = R, records per KV head
    const char* plan;    // [R+1][PLAN_BYTES]
    int* sched;          // [2] = {ticket counter, workgroups done}; both 0 between launches
    // fused paged append (optional): rows whose plan offset has bit 63 set are read from k_new / v_new
    // (offset = row index * new_st * 2 bytes) and workgroup b < n_new also copies row b into the pool
    const _Float16* k_new;
    const _Float16* v_new;
    const int32_t* cache_loc;
    int64_t new_st;
    int n_new;
    unsigned long long* dbg;  // internal: per-phase s_memtime stamps [workgroup][16 tiles][8], or null
};

// DB = false: two workgroups per CU, one K and one V stage each (K(i+1) streams in after tile i's QK^T, V(i+1)
//              after its PV), Q staged in the idle P buffer.
// DB = true : one workgroup per CU, two K and two V stages, loads run TWO tiles ahead: K(i+2) is issued when
//             tile i's QK^T is done, V(i+2) when its PV is done, so tile i+1 has had a whole tile's time to
//             land (a 64 KB burst takes ~3 us under load; with one tile in flight the CU idles on it).
//             Q in its own buffer, three plan records resident.  Static unit split only.
template <int D, bool DB>
struct StreamSmem {
    static constexpr int STAGE = TILE * D * 2;  // one K (or V) tile
    static constexpr int NBUF = DB ? 2 : 1;
    static constexpr int K_OFF = 0;
    static constexpr int V_OFF = K_OFF + NBUF * STAGE;
    static constexpr int P_OFF = V_OFF + NBUF * STAGE;
    static constexpr int Q_OFF = DB ? P_OFF + MQ * TILE * 2 : P_OFF;
    static constexpr int NMETA = DB ? 3 : 2;                // plan records resident in LDS
    static constexpr int META_OFF = Q_OFF + MQ * TILE * 2;
    static constexpr int WMAX_OFF = META_OFF + NMETA * PLAN_BYTES;
    static constexpr int WSUM_OFF = WMAX_OFF + 4 * MQ * 4;
    static constexpr int UNIT_OFF = WSUM_OFF + 4 * MQ * 4;  // int[4] ring of unit ids, -1 = end of stream
    static constexpr int BYTES = UNIT_OFF + 16;
};

// 64 lanes x 16 bytes, global (per-lane address) -> LDS (lds_dst + 16*lane).  M0 is not
// otherwise used by this kernel (checked in the .s), so it is written, not saved.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int D, bool DB>
__global__ __launch_bounds__(512, DB ? 2 : 4) void stage1_stream_kernel(StreamParams sp) {
    constexpr int CH = D / 8;
    constexpr int KS = D / 16;
    constexpr int DPT = 32 * CH / 64;  // DMA instructions per 32-key slice of a K (or V) tile
    constexpr int LPT = DPT;           // per loader wave: one slice (4 loaders)
    static_assert(D == 128, "streaming stage 1 is instantiated for head_dim 128");
    using SM = StreamSmem<D, DB>;
    const Stage1Params& p = sp.s;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* sP = reinterpret_cast<_Float16*>(smem + SM::P_OFF);
    float* sWmax = reinterpret_cast<float*>(smem + SM::WMAX_OFF);
    float* sWsum = reinterpret_cast<float*>(smem + SM::WSUM_OFF);

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;

    // Work distribution.  Position i of a workgroup's unit stream is known two tiles ahead: lane 0 of compute
    // wave 0 publishes unit(i+2) in the sUnit ring before barrier H of tile i.
    //  * default: balanced static split of the head-major unit sequence, no atomics;
    //  * experimental (DEFT_STAGE1_ABLATE bit 32): guided chunks from one atomic ticket counter — 60 % of
    //    every head's units in chunks of 4, 25 % in chunks of 2, 15 % one by one — meant to even out the
    //    workgroups' finish times (equal static shares finish +-20 % apart: a 9 us tail on a 29 us bulk).
    //    Measured 4-6 us SLOWER than the static split on the 4k x 32 tree: the returning atomic of a compute
    //    wave queues behind ~100 KB of loader DMA in the CU's memory FIFO, and every chunk is its own group
    //    (own Q fetch, own partial).  Kept for the next round (ticket prefetch two tiles ahead).
    const int bid = blockIdx.x;
    const int RH = sp.hdr[0];  // records per KV head (written by the plan kernels)
    const int U = RH * p.Hkv;  // units, head-major
    const int W = (int)gridDim.x;
    const int per = U / W, rem = U - per * W;
    const bool guided = !DB && per >= 4 && (p.ablate & 32);  // off by default: measured slower, see below
    const int n_static = per + (bid < rem ? 1 : 0);
    const int my_base = bid * per + min(bid, rem);
    // chunks are laid out per KV head and tickets walk the heads round-robin (ticket k -> head k % Hkv,
    // chunk k / Hkv of that head): at any moment the active chunks cover all heads evenly.  Handing out the
    // head-major sequence in order instead makes every workgroup read the same few heads at the same time,
    // i.e. the same 256-byte column of every token row, which serialises on a few HBM channels (-10 %).
    const int R1 = guided ? (int)(0.60f * RH) / 4 * 4 : 0;
    const int R2 = guided ? R1 + (int)(0.25f * RH) / 2 * 2 : 0;
    const int n1 = R1 / 4, n2 = (R2 - R1) / 2, n3 = RH - R2;
    auto chunk_of = [&](int k, int& cs, int& ce) {
        const int head = k % p.Hkv, ci = k / p.Hkv;
        int st, len;
        if (ci < n1) { st = 4 * ci; len = 4; }
        else if (ci < n1 + n2) { st = R1 + 2 * (ci - n1); len = 2; }
        else if (ci < n1 + n2 + n3) { st = R2 + (ci - n1 - n2); len = 1; }
        else { cs = -1; ce = -1; return; }
        cs = head * RH + st;
        ce = cs + len;
    };
    int* sUnit = reinterpret_cast<int*>(smem + SM::UNIT_OFF);
    int pub_u = -1, pub_end = -1;  // publisher state (lane 0 of compute wave 0): last published unit, end of its chunk
    if (tid == 0) {
        int a, b2;
        if (guided) {
            int cs, ce;
            chunk_of(atomicAdd(sp.sched, 1), cs, ce);
            a = cs;
            if (ce - cs >= 2) {
                b2 = cs + 1;
            } else if (cs >= 0) {
                chunk_of(atomicAdd(sp.sched, 1), cs, ce);
                b2 = cs;
            } else {
                b2 = -1;
            }
            pub_u = b2;
            pub_end = ce;
        } else {
            a = n_static > 0 ? my_base : -1;
            b2 = n_static > 1 ? my_base + 1 : -1;
        }
        sUnit[0] = a;
        sUnit[1] = a < 0 ? -1 : b2;
    }

    // ---- loop-invariant lane constants -------------------------------------------------
    // DMA: instruction i of a tile stages keys 32w + 4i + (l>>4), LDS chunk position l&15.
    //   K source chunk = pos ^ (key & 15) = (pos ^ (l>>4)) ^ 4*(i&3);  V source chunk = pos ^ 4*(key&3)
    const int dpos = l & 15, dkey = l >> 4;
    int kchunk_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk_b[j] = ((dpos ^ dkey) ^ (4 * j)) * 16;
    const int vchunk_b = (dpos ^ (4 * (dkey & 3))) * 16;
    const bool is_loader = w >= 4;  // waves 4..7 stream; waves 0..3 compute
    const int lw = w - 4;           // loader lw stages keys [32 lw, 32 lw + 32)
    const uint32_t ldsK = SM::K_OFF + (uint32_t)(lw < 0 ? 0 : lw) * 32u * D * 2u;
    const uint32_t ldsV = SM::V_OFF + (uint32_t)(lw < 0 ? 0 : lw) * 32u * D * 2u;
    // S^T A fragments: row 32w + c, chunk (2ks + h) ^ (c & 15)  ->  byte (((h ^ c) & 15) * 16) ^ (32 * ks)
    const int krow_b = (32 * w + c) * D * 2;
    const int kcol_b = ((h ^ c) & 15) * 16;
    // O^T A fragments (transpose reads): lane (g = l>>4, x = l&15) supplies 8 bytes of row
    //   16ks + 8(g>>1) + (x>>2) [+4], d = 32w + 16(g&1) + 4(x&3), chunk XOR-ed by 4*(row & 3) = 4*(x>>2)
    const int tg = l >> 4, tx = l & 15;
    const int vtr_b = SM::V_OFF + (8 * (tg >> 1) + (tx >> 2)) * D * 2 +
                      (((4 * w + 2 * (tg & 1) + ((tx & 3) >> 1)) ^ (4 * (tx >> 2))) * 16) + (tx & 1) * 8;
    // P: row c, 16-byte chunks XOR-ed by (c & 15)
    const int prow_b = SM::P_OFF + c * TILE * 2;

    auto meta = [&](int b) { return smem + SM::META_OFF + b * PLAN_BYTES; };
    auto issue_meta = [&](int tt, int b) {  // one 1 KB DMA by each of loaders 0 and 1
        const char* rec = sp.plan + (int64_t)tt * PLAN_BYTES;
        if (lw < 2) dma16(rec + 1024 * lw + 16 * l, SM::META_OFF + (uint32_t)b * PLAN_BYTES + 1024u * (uint32_t)lw);
    };
    int64_t rowoff[LPT];  // pool byte offsets of this loader lane's LPT rows of the NEXT tile (set at D, reused at I)
    auto load_rowoff = [&](int b) {
        const int64_t* ro = reinterpret_cast<const int64_t*>(meta(b) + PLAN_ROWOFF);
#pragma unroll
        for (int i = 0; i < LPT; ++i) rowoff[i] = ro[32 * lw + 4 * i + dkey];
    };
    constexpr int64_t NEW_ROW = (int64_t)1 << 63;  // plan offset flag: row lives in k_new / v_new
    auto issue_k = [&](int head, int kb) {
        const char* hb = reinterpret_cast<const char*>(p.k) + (int64_t)head * p.kv_sh * 2;
        const char* nb = reinterpret_cast<const char*>(sp.k_new) + (int64_t)head * D * 2;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const char* src = rowoff[i] < 0 ? nb + (rowoff[i] & ~NEW_ROW) : hb + rowoff[i];
            dma16(src + kchunk_b[i & 3], ldsK + (uint32_t)kb * SM::STAGE + (uint32_t)i * 1024u);
        }
    };
    auto issue_v = [&](int head, int kb) {
        const char* hb = reinterpret_cast<const char*>(p.v) + (int64_t)head * p.kv_sh * 2 + vchunk_b;
        const char* nb = reinterpret_cast<const char*>(sp.v_new) + (int64_t)head * D * 2 + vchunk_b;
#pragma unroll
        for (int i = 0; i < LPT; ++i)
            dma16(rowoff[i] < 0 ? nb + (rowoff[i] & ~NEW_ROW) : hb + rowoff[i], ldsV + (uint32_t)kb * SM::STAGE + (uint32_t)i * 1024u);
    };
    auto read_desc = [&](int b) {  // {cnt, prow, run_start, len}, wave-uniform
        intx4 d = *reinterpret_cast<const intx4*>(meta(b) + PLAN_DESC);
        intx4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(d[j]);
        return r;
    };
    auto stamp = [&](int i, int k) {
        if (sp.dbg && tid == 0 && i < 16) sp.dbg[((int64_t)bid * 16 + i) * 8 + k] = __builtin_amdgcn_s_memtime();
    };

    // Q rows of the group that opens at a tile: the loaders stage them in the P buffer, which is idle
    // between a tile's PV reads (barrier H) and the next tile's P writes (after barrier C).
    // Layout like K: row c, 16-byte chunks XOR-ed by (c & 15).  Rows beyond n_vrows alias the first
    // query vector of the group (their mask bits are 0).  2 DMA instructions per loader.
    auto issue_q = [&](int b, int head) {
        const int32_t* qs = reinterpret_cast<const int32_t*>(meta(b) + PLAN_QSRC);
        const char* hb = reinterpret_cast<const char*>(p.q) + (int64_t)head * p.G * p.q_sh * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * lw + 4 * i + dkey;
            const int chunk = dpos ^ (row & 15);
            dma16(hb + (int64_t)qs[row] * 2 + chunk * 16, SM::Q_OFF + (uint32_t)(8 * lw + 4 * i) * 256u);
        }
    };

    if (sp.dbg && tid == 0) sp.dbg[((int64_t)bid * 16 + 15) * 8 + 6] = wall_clock64();
    auto finish = [&]() {  // the last workgroup to leave re-arms the scheduler words for the next launch
        if (tid == 0) {
            if (sp.dbg) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                sp.dbg[((int64_t)bid * 16 + 15) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
            }
            if (sp.dbg) sp.dbg[((int64_t)bid * 16 + 15) * 8 + 7] = wall_clock64();
            if (atomicAdd(sp.sched + 1, 1) == (int)gridDim.x - 1) {
                sp.sched[0] = 0;
                sp.sched[1] = 0;
            }
        }
    };
    // ---- fused paged append: workgroup b writes new-token row b into the pool (nobody reads those
    //      pool rows in this launch: the loaders take them from k_new / v_new) -------------------
    // (the copy jobs go to the LAST workgroups of the grid: with a balanced static split those own one
    //  unit less than the first ones whenever the units do not divide evenly)
    const int copy_job = (int)gridDim.x - 1 - bid;
    if (copy_job < sp.n_new) {
        const int64_t dst = (int64_t)sp.cache_loc[copy_job] * p.kv_ss;
        const int chunks = p.Hkv * (D / 8);  // 16-byte pieces per K (or V) row
        for (int i = tid; i < chunks; i += blockDim.x) {
            const int hd = i / (D / 8), ch = i - hd * (D / 8);
            const int64_t so = (int64_t)copy_job * sp.new_st + hd * D + ch * 8;
            const int64_t d_o = dst + (int64_t)hd * p.kv_sh + ch * 8;
            const uintx4 kk = *reinterpret_cast<const uintx4*>(sp.k_new + so);
            const uintx4 vv = *reinterpret_cast<const uintx4*>(sp.v_new + so);
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.k) + d_o) = kk;
            *reinterpret_cast<uintx4*>(const_cast<_Float16*>(p.v) + d_o) = vv;
        }
    }
    // ---- prologue -----------------------------------------------------------------
    lds_barrier();  // sUnit[0..1] visible
    int ucur = __builtin_amdgcn_readfirstlane(sUnit[0]);
    int unext = __builtin_amdgcn_readfirstlane(sUnit[1]);
    if (ucur < 0) {
        finish();
        return;
    }
    int kvh = ucur / RH;
    int t = ucur - kvh * RH;  // record index within the head
    if constexpr (DB) {
        // ======================= two tiles ahead, static units u(i) = my_base + i =======================
        auto exists = [&](int i) { return i < n_static; };
        auto head_of = [&](int i) { return (my_base + i) / RH; };
        auto rec_of = [&](int i) { return (my_base + i) % RH; };
        if (is_loader) {
            issue_meta(rec_of(0), 0);
            if (exists(1)) issue_meta(rec_of(1), 1);
            if (exists(2)) issue_meta(rec_of(2), 2);
            wait_vm<0>();
        }
        lds_barrier();
        if (is_loader) {
            load_rowoff(0);
            issue_q(0, head_of(0));  // the first tile of a stream always opens a group
            issue_k(head_of(0), 0);
            issue_v(head_of(0), 0);
            if (exists(1)) {
                load_rowoff(1);
                issue_k(head_of(1), 1);
                issue_v(head_of(1), 1);
            }
            for (int i = 0; i < n_static; ++i) {
                // A(i): everything of tile i (and the plan records up to i+2) has landed; tile i+1 may be in flight
                if (exists(i + 1)) wait_vm<2 * LPT>(); else wait_vm<0>();
                lds_barrier();
                lds_barrier();  // C(i): QK^T(i) done (K stage i&1 free), Q(i) in registers, masks(i) consumed
                if (exists(i + 1)) {
                    const bool next_opens = (rec_of(i + 1) == 0) || (read_desc((i + 1) % 3)[2] != 0);
                    if (next_opens) issue_q((i + 1) % 3, head_of(i + 1));
                }
                if (exists(i + 3)) issue_meta(rec_of(i + 3), i % 3);
                if (exists(i + 2)) {
                    load_rowoff((i + 2) % 3);
                    issue_k(head_of(i + 2), i & 1);
                }
                lds_barrier();  // F(i)
                lds_barrier();  // H(i): PV(i) done (V stage i&1 free)
                if (exists(i + 2)) issue_v(head_of(i + 2), i & 1);
            }
            return;
        }
    } else {
        if (is_loader) {
            issue_meta(t, 0);
            wait_vm<0>();
        }
        lds_barrier();
        if (is_loader) {
            load_rowoff(0);
            issue_k(kvh, 0);
            issue_q(0, kvh);  // the first tile of a stream always opens a group
            if (unext >= 0) issue_meta(unext % RH, 1);
            issue_v(kvh, 0);

            // ---- loader loop: same barrier sequence as the compute waves below ---------------
            for (int i = 0;; ++i) {
                const int mb = i & 1;
                const bool last = unext < 0;
                const int kvh_next = last ? 0 : unext / RH;
                const int t_next = last ? 0 : unext - kvh_next * RH;
                wait_vm<LPT>();  // A: K(u) and plan record (u+1) landed
                lds_barrier();
                lds_barrier();   // C: compute waves are done with sK
                if (!last) {
                    load_rowoff(mb ^ 1);
                    issue_k(kvh_next, 0);
                }
                if (last) wait_vm<0>(); else wait_vm<LPT>();  // F: V(u) landed
                lds_barrier();
                lds_barrier();   // H: compute waves are done with sV and sP; unit(i+2) published
                if (last) break;
                const int u2 = __builtin_amdgcn_readfirstlane(sUnit[(i + 2) & 3]);
                const bool next_opens = (unext != ucur + 1) || (t_next == 0) || (read_desc(mb ^ 1)[2] != 0);
                if (next_opens) issue_q(mb ^ 1, kvh_next);
                if (u2 >= 0) issue_meta(u2 % RH, mb);
                issue_v(kvh_next, 0);
                ucur = unext;
                unext = u2;
                t = t_next;
                kvh = kvh_next;
            }
            return;
        }
    }

    // running state of the current group (query row c of this lane)
    half8 qf[KS];
    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    int g_orow = 0;  // this lane's partial row of the open group, relative to kvh * G * rows
    bool qvalid = false;

    bool prev_adjacent = false;  // unit(i) == unit(i-1) + 1
    for (int i = 0;; ++i) {
        const int mb = DB ? i % 3 : (i & 1);                    // plan record slot of this unit
        const int mn = DB ? (i + 1) % 3 : ((i & 1) ^ 1);        // ... of the next unit
        const int stg = DB ? (i & 1) : 0;                       // K/V stage of this unit
        const bool last = unext < 0;
        const intx4 cur = read_desc(mb);
        const int orow_c = reinterpret_cast<const int32_t*>(meta(mb) + PLAN_OROW)[c];
        const bool g_start = !prev_adjacent || (t == 0) || (cur[2] != 0);

        // ---- A: K(u) and plan record (u+1) landed ------------------------------------
        stamp(i, 0);
        lds_barrier();
        stamp(i, 1);
        if (g_start) {  // new group: its Q rows sit in the P buffer (staged by the loaders)
            g_orow = orow_c;
            qvalid = c < cur[0];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[ks] = *reinterpret_cast<const half8*>(smem + SM::Q_OFF + c * TILE * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
            m_run = -INFINITY;
            l_run = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
        }
        const bool next_adjacent = !last && (unext == ucur + 1);
        bool g_end = true;  // plan record (u+1) is visible now
        if (next_adjacent) g_end = (t + 1 == RH) || (read_desc(mn)[2] != 0);
        // a ticket for position i+2 (when it opens a new chunk) travels while this tile computes; compute
        // waves issue no other loads, so the wait before barrier H costs nothing
        int ticket = 0;
        const bool need_ticket = guided && !last && (pub_u + 1 >= pub_end);
        if (tid == 0 && need_ticket)  // asm: hipcc would otherwise wait for the returned value right here
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(sp.sched), "v"(1) : "memory");

        // ---- B: S^T for this wave's 32 keys, scale, mask, row max --------------------
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 a = *reinterpret_cast<const half8*>(smem + SM::K_OFF + stg * SM::STAGE + krow_b + (kcol_b ^ (32 * ks)));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], acc, 0, 0, 0);
        }
        float s[16];
        float mx = -INFINITY;
        {
            const uint32_t* masks = reinterpret_cast<const uint32_t*>(meta(mb) + PLAN_MASK);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uintx4 m4 = *reinterpret_cast<const uintx4*>(masks + 32 * w + 8 * g4 + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g4 + j;
                    s[r] = ((m4[j] >> c) & 1u) ? acc[r] * p.scale_log2e : -INFINITY;
                    mx = fmaxf(mx, s[r]);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (h == 0) sWmax[w * MQ + c] = mx;

        // ---- C: every wave is done with sK; tile maxima visible -------------------------
        stamp(i, 2);
        lds_barrier();
        stamp(i, 3);

        // ---- D: (loaders) K(u+1) streams in while this tile's softmax and PV run ------------

        // ---- E: online softmax update, P (fp16) -> LDS -----------------------------------
        const float m_tile = fmaxf(fmaxf(sWmax[c], sWmax[MQ + c]), fmaxf(sWmax[2 * MQ + c], sWmax[3 * MQ + c]));
        const float m_new = fmaxf(m_run, m_tile);
        const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
        float sum = 0.f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            half4 p4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const _Float16 ph = (_Float16)__builtin_amdgcn_exp2f(s[4 * g4 + j] - msafe);
                p4[j] = ph;
                sum += (float)ph;
            }
            const int pos = (4 * w + g4) ^ (c & 15);
            *reinterpret_cast<half4*>(sP + c * TILE + pos * 8 + 4 * h) = p4;
        }
        sum += __shfl_xor(sum, 32);
        if (h == 0) sWsum[w * MQ + c] = sum;

        // ---- F: V(u) landed; P and row sums visible ----------------------------------------
        stamp(i, 4);
        lds_barrier();
        stamp(i, 5);

        // ---- G: O^T = alpha * O^T + V^T P^T, 32 output columns per wave --------------------
        {
            const float tile_sum = sWsum[c] + sWsum[MQ + c] + sWsum[2 * MQ + c] + sWsum[3 * MQ + c];
            l_run = l_run * alpha + tile_sum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < TILE / 16; ++ks) {
                typedef __attribute__((address_space(3))) short4v* lds_s4;
                const int vb = vtr_b + stg * SM::STAGE;
                const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + (16 * ks) * D * 2));
                const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + (16 * ks + 4) * D * 2));
                union {
                    short4v s4[2];
                    half8 h8;
                } av;
                av.s4[0] = a0;
                av.s4[1] = a1;
                const half8 b = *reinterpret_cast<const half8*>(smem + prow_b + (((2 * ks + h) ^ (c & 15)) * 16));
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(av.h8, b, o, 0, 0, 0);
            }
        }

        // ---- H: every wave is done with sV and sP; unit(i+2) published -------------------------
        stamp(i, 6);
        if (!DB && tid == 0 && !last) {
            int u2;
            if (!guided) {
                u2 = (i + 2 < n_static) ? my_base + i + 2 : -1;
            } else if (need_ticket) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket)::"memory");
                chunk_of(ticket, u2, pub_end);
                pub_u = u2;
            } else {
                u2 = ++pub_u;
            }
            sUnit[(i + 2) & 3] = u2;
        }
        lds_barrier();
        stamp(i, 7);

        // ---- I: (loaders) plan record (u+2) and V(u+1) stream in during the next tile's QK^T / softmax

        // ---- J: bookkeeping for the merge; partial out at the end of a group --------------------
        const int64_t head_rows = (int64_t)kvh * p.G * p.rows;
        if (!(p.ablate & 8) && !g_start && qvalid && w == 0 && h == 0)
            p.partial_lse[head_rows + orow_c] = -INFINITY;  // row folded into its group's partial
        if (!(p.ablate & 16) && g_end && qvalid) {
            const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
            const int64_t prow_idx = head_rows + g_orow;
            float* po = p.partial_o + prow_idx * D + 32 * w + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                floatx4 v4 = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
                *reinterpret_cast<floatx4*>(po + 8 * g4) = v4;
            }
            if (w == 0 && h == 0)
                p.partial_lse[prow_idx] = (l_run > 0.f) ? (m_run + __builtin_amdgcn_logf(l_run)) * LN2 : -INFINITY;
        }
        if (last) break;
        prev_adjacent = next_adjacent;
        ucur = unext;
        if constexpr (DB)
            unext = (i + 2 < n_static) ? my_base + i + 2 : -1;
        else
            unext = __builtin_amdgcn_readfirstlane(sUnit[(i + 2) & 3]);
        kvh = ucur / RH;
        t = ucur - kvh * RH;
    }
    finish();
}

// ---------------------------------------------------------------------------
// Plan kernels (once per decode step): metadata -> unit list -> records
// ---------------------------------------------------------------------------
// Byte offset of a pool slot's row, or (bit 63 | offset into k_new / v_new) when the slot is one of
// this step's new tokens and the caller uses the fused append.
__device__ __forceinline__ int64_t plan_rowoff(int64_t slot, int64_t kv_stride_slot, const int32_t* cache_loc, int n_new,
                                               int64_t new_row_bytes) {
    for (int i = 0; i < n_new; ++i)
        if ((int64_t)cache_loc[i] == slot) return ((int64_t)1 << 63) | ((int64_t)i * new_row_bytes);
    return slot * kv_stride_slot * 2;  // fp16 bytes
}

struct UnitList {   // all int32, capacity `cap` each
    int32_t* src;   // Flatten: block index; Node: entry index
    int32_t* aux;   // Flatten: 0;           Node: 128-slot tile index within the entry
    int32_t* pass;  // 32-row pass of the unit's virtual query rows
    int32_t* flags; // bit 0: opens a run (query list differs from the previous unit's)
    int32_t* prow;  // first partial row of the unit's tile
};

// Flatten: one workgroup.  Phase 1 (parallel over blocks): does block t open a run, how many passes.
// Phase 2 (one thread): emit units run by run, pass-major inside a run so that consecutive units fold.
__global__ __launch_bounds__(256) void flatten_units_kernel(const int64_t* block_q, const int64_t* block_q_cnts,
                                                            const int64_t* block_q_offset, int NB, int G, int cap,
                                                            UnitList ul, int32_t* hdr, int32_t* sched) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sOpen = reinterpret_cast<int*>(smem);  // [NB]
    int* sPass = sOpen + NB;                    // [NB]
    for (int t = threadIdx.x; t < NB; t += blockDim.x) {
        const int cnt = (int)block_q_cnts[t];
        bool open = (t == 0);
        if (t > 0) {
            open = cnt != (int)block_q_cnts[t - 1];
            const int64_t a = block_q_offset[t], b = block_q_offset[t - 1];
            for (int i = 0; !open && i < cnt; ++i) open = block_q[a + i] != block_q[b + i];
        }
        sOpen[t] = open ? 1 : 0;
        sPass[t] = (cnt * G + MQ - 1) / MQ;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = 0;
        for (int ta = 0; ta < NB;) {
            int tb = ta + 1;
            while (tb < NB && !sOpen[tb]) ++tb;
            const int np = sPass[ta];
            for (int ps = 0; ps < np; ++ps)
                for (int t = ta; t < tb && r < cap; ++t, ++r) {
                    ul.src[r] = t;
                    ul.aux[r] = 0;
                    ul.pass[r] = ps;
                    ul.flags[r] = (t == ta) ? 1 : 0;
                    ul.prow[r] = (int)block_q_offset[t];
                }
            ta = tb;
        }
        hdr[0] = r;
        sched[0] = 0;
        sched[1] = 0;
    }
}

// One workgroup of 128 threads per unit (+ the sentinel): pack its record.
__global__ __launch_bounds__(128) void flatten_records_kernel(const int64_t* block_q, const int64_t* block_q_cnts,
                                                              const int64_t* block_bitmasks, const int64_t* block_kv,
                                                              const int64_t* block_lens, int G, int rows, int64_t q_st,
                                                              int64_t q_sh, int64_t kv_stride_slot, UnitList ul,
                                                              const int32_t* hdr, char* plan, int32_t* row_q,
                                                              const int32_t* cache_loc, int n_new, int64_t new_row_bytes) {
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    const int R = hdr[0];
    if (r > R) return;
    char* rec = plan + (int64_t)r * PLAN_BYTES;
    int64_t* ro = reinterpret_cast<int64_t*>(rec + PLAN_ROWOFF);
    uint32_t* mk = reinterpret_cast<uint32_t*>(rec + PLAN_MASK);
    int32_t* desc = reinterpret_cast<int32_t*>(rec + PLAN_DESC);
    if (r == R) {  // sentinel
        ro[k] = 0;
        mk[k] = 0u;
        if (k == 0) {
            desc[0] = 0;
            desc[1] = 0;
            desc[2] = 1;
            desc[3] = 0;
        }
        return;
    }
    const int t = ul.src[r];
    const int ps = ul.pass[r];
    const int prow = ul.prow[r];
    const int len = (int)block_lens[t];
    const int cnt = (int)block_q_cnts[t];
    const int nv = min(MQ, cnt * G - MQ * ps);  // virtual rows of this pass
    const bool live = k < len;
    ro[k] = plan_rowoff(block_kv[(int64_t)t * TILE + (live ? k : 0)], kv_stride_slot, cache_loc, n_new, new_row_bytes);
    const uint32_t qmask = live ? (uint32_t)block_bitmasks[(int64_t)t * TILE + k] : 0u;
    uint32_t vm = 0u;
    for (int v = 0; v < nv; ++v) vm |= ((qmask >> ((MQ * ps + v) / G)) & 1u) << v;
    mk[k] = vm;
    if (k < MQ) {
        int qs = 0, orow = 0;
        if (k < nv) {
            const int qi = (MQ * ps + k) / G, g = (MQ * ps + k) % G;
            qs = (int)(block_q[prow + qi] * q_st + g * q_sh);
            orow = g * rows + prow + qi;
        } else if (nv > 0) {
            qs = (int)(block_q[prow + (MQ * ps) / G] * q_st + ((MQ * ps) % G) * q_sh);  // alias a real row
        }
        reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = qs;
        reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = orow;
        if (ps == 0 && k < cnt) row_q[prow + k] = (int32_t)block_q[prow + k];
    }
    if (k == 0) {
        desc[0] = nv;
        desc[1] = prow;
        desc[2] = ul.flags[r] & 1;
        desc[3] = len;
    }
}

// Node mode (tree_attention.py:14-293): every entry (a node's KV x up to 32 of its queries) is cut into
// 128-slot tiles; all live slots are visible to all of the entry's queries.  Consecutive tiles of one
// entry fold, which is the reference's serial online-softmax walk (:230-276) without the serialisation.
__global__ __launch_bounds__(256) void node_units_kernel(const int64_t* node_kv_len, const int64_t* node_q_len, int NE, int G,
                                                         int cap, int64_t rows_cap, UnitList ul, int32_t* hdr,
                                                         int32_t* sched, int32_t* row_q) {
    for (int64_t i = threadIdx.x; i < rows_cap; i += blockDim.x) row_q[i] = -1;
    if (threadIdx.x == 0) {
        int r = 0, rowbase = 0;
        for (int e = 0; e < NE; ++e) {
            const int nt = (int)((node_kv_len[e] + TILE - 1) / TILE);
            const int ql = (int)node_q_len[e];
            const int np = (ql * G + MQ - 1) / MQ;
            for (int ps = 0; ps < np; ++ps)
                for (int tt = 0; tt < nt && r < cap; ++tt, ++r) {
                    ul.src[r] = e;
                    ul.aux[r] = tt;
                    ul.pass[r] = ps;
                    ul.flags[r] = (tt == 0) ? 1 : 0;
                    ul.prow[r] = rowbase + tt * ql;
                }
            rowbase += nt * ql;
        }
        hdr[0] = r;
        sched[0] = 0;
        sched[1] = 0;
    }
}

__global__ __launch_bounds__(128) void node_records_kernel(const int64_t* node_kv, const int64_t* node_kv_offset,
                                                           const int64_t* node_kv_len, const int64_t* node_q,
                                                           const int64_t* node_q_offset, const int64_t* node_q_len, int G,
                                                           int rows, int64_t q_st, int64_t q_sh, int64_t kv_stride_slot,
                                                           UnitList ul, const int32_t* hdr, char* plan, int32_t* row_q,
                                                           const int32_t* cache_loc, int n_new, int64_t new_row_bytes) {
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    const int R = hdr[0];
    if (r > R) return;
    char* rec = plan + (int64_t)r * PLAN_BYTES;
    int64_t* ro = reinterpret_cast<int64_t*>(rec + PLAN_ROWOFF);
    uint32_t* mk = reinterpret_cast<uint32_t*>(rec + PLAN_MASK);
    int32_t* desc = reinterpret_cast<int32_t*>(rec + PLAN_DESC);
    if (r == R) {  // sentinel
        ro[k] = 0;
        mk[k] = 0u;
        if (k == 0) {
            desc[0] = 0;
            desc[1] = 0;
            desc[2] = 1;
            desc[3] = 0;
        }
        return;
    }
    const int e = ul.src[r], tt = ul.aux[r], ps = ul.pass[r], prow = ul.prow[r];
    const int64_t kv0 = node_kv_offset[e] + (int64_t)tt * TILE;
    const int len = (int)min((int64_t)TILE, node_kv_len[e] - (int64_t)tt * TILE);
    const int64_t q0 = node_q_offset[e];
    const int ql = (int)node_q_len[e];
    const int nv = min(MQ, ql * G - MQ * ps);
    const bool live = k < len;
    ro[k] = plan_rowoff(node_kv[kv0 + (live ? k : 0)], kv_stride_slot, cache_loc, n_new, new_row_bytes);
    mk[k] = live ? (nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u)) : 0u;
    if (k < MQ) {
        int qs = 0, orow = 0;
        if (k < nv) {
            const int qi = (MQ * ps + k) / G, g = (MQ * ps + k) % G;
            qs = (int)(node_q[q0 + qi] * q_st + g * q_sh);
            orow = g * rows + prow + qi;
        } else if (nv > 0) {
            qs = (int)(node_q[q0 + (MQ * ps) / G] * q_st + ((MQ * ps) % G) * q_sh);
        }
        reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = qs;
        reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = orow;
        if (ps == 0 && k < ql) row_q[prow + k] = (int32_t)node_q[q0 + k];
    }
    if (k == 0) {
        desc[0] = nv;
        desc[1] = prow;
        desc[2] = ul.flags[r] & 1;
        desc[3] = len;
    }
}

}  // namespace deft
