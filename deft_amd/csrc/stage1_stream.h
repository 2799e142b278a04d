// Stage 1 (Flatten and Node), streaming form: persistent workgroups, LDS-DMA pipeline.
//
// Included by deft_kernels.hip (needs its typedefs and Stage1Params).
//
//   * plan kernels (plan_kernels.h, once per decode step) pack each 128-slot KV tile's metadata into one 2 KB record:
//     the byte offset of every KV row in the pool, a 32-bit virtual-query mask per slot and the
//     query / partial-row maps (PLAN_* below);
//   * ONE workgroup per CU = 4 compute waves + 4 LOADER waves.  Only the loaders issue DMA, so the
//     compute waves never sit in the CU's vector-memory FIFO (measured: with compute waves issuing
//     their own DMA, half of every tile's time was spent blocked in the issue of 16 instructions);
//   * K and V tiles travel HBM -> LDS by `global_load_lds_dwordx4` (no VGPR round trip), double
//     buffered, TWO tiles ahead of the compute waves: K(i+2) is issued when tile i's QK^T is done,
//     V(i+2) when its PV is done.  Plan records and the Q rows of a group ride the same queue;
//   * V^T MFMA fragments come from `ds_read_b64_tr_b16`; V is stored with its 16-byte chunks XOR-ed
//     by 4*(key&3), K with chunks XOR-ed by key&15 for the row-per-lane b128 reads;
//   * tiles of one (KV head, run) that a workgroup meets back to back are folded with an online
//     softmax in registers and emit ONE partial;
//   * work distribution: the first rounds are a static INTERLEAVED walk (unit i of workgroup b is
//     chunk i*W + b of the tile-major / head-fastest sequence: the whole chip advances through the
//     pool as one front, and with W % Hkv == 0 a workgroup stays on one head so prefix tiles fold);
//     the last ~30 % of the units are handed out one by one from an atomic ticket counter so that
//     workgroups finish together (static shares finish 9 us apart on a 30 us kernel: ramp-up skew
//     and 10-vs-11-tile shares).  The ticket atomic is issued by a LOADER wave a tile before it is
//     needed and retires under that wave's normal counted wait, so it never stalls anyone.
//
// All DMA issue and all waits on it are inline asm: hipcc neither counts asm VMEM operations nor
// drains them at a raw s_barrier, which is what lets loads stay in flight across barriers
// (cdna_hip_programming.md §5.7).  Per loader wave, LPT = its DMA instructions per K or V tile:
//   issue order  ... | [ticket] Q(i+1) rec(i+3) K(i+2) | V(i+2) | ...      (after C(i) | after H(i))
//   "tile i+1 and everything older landed"  <=>  at most K(i+2), V(i+2) outstanding -> vmcnt(2 LPT)
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
constexpr int PLAN_DESC = 1536;  // int32[8]    n_vrows, prow, opens_run, run_id, chunk tiles (0 = follower), first follower record, -, -
constexpr int PLAN_QSRC = 1600;  // int32[32]   element offset of row v's Q vector from q + kvh*G*q_stride_head
constexpr int PLAN_OROW = 1728;  // int32[32]   partial row of row v, relative to kvh*G*rows: g*rows + prow + qi
constexpr int PLAN_HDR = 4096;   // plan header: int32 R (records per head) at +0, done counter at +64, ticket counters
constexpr int NTICKET = 8;       // ... NTICKET of them at +512 + 256 k: sched[112 + 64 k] (sched = header + 64 bytes)
__host__ __device__ inline int ticket_word(int k) { return 112 + 64 * k; }

struct StreamParams {
    Stage1Params s;
    const int32_t* hdr;  // plan header: hdr[0] = R, records per KV head
    const char* plan;    // [R+1][PLAN_BYTES]
    int cap;             // records the plan buffer holds (>= R + 1): bound for speculative record prefetch
    int dyn_pct;         // share (%) of a workgroup's units handed out by ticket at the end of the walk; 0 = static
    int* sched;          // sched[0] = workgroups done, sched[ticket_word(k)] = ticket counter k; all 0 between launches
    // fused paged append (optional): rows whose plan offset has bit 63 set are read from k_new / v_new
    // (offset = row index * new_st * 2 bytes) and workgroup b < n_new also copies row b into the pool
    const _Float16* k_new;
    const _Float16* v_new;
    const int32_t* cache_loc;
    int64_t new_st;
    int n_new;
    unsigned long long* dbg;  // internal: per-phase s_memtime stamps [workgroup][16 tiles][8], or null
};

template <int D>
struct StreamSmem {
    static constexpr int STAGE = TILE * D * 2;  // one K (or V) tile
    static constexpr int K_OFF = 0;             // two K stages
    static constexpr int V_OFF = K_OFF + 2 * STAGE;
    static constexpr int P_OFF = V_OFF + 2 * STAGE;         // P^T fp16 [32][128]
    static constexpr int Q_OFF = P_OFF + MQ * TILE * 2;     // Q rows of the opening group [32][D]
    static constexpr int NMETA = 3;                         // plan records resident in LDS
    static constexpr int META_OFF = Q_OFF + MQ * TILE * 2;
    static constexpr int WMAX_OFF = META_OFF + NMETA * PLAN_BYTES;
    static constexpr int WSUM_OFF = WMAX_OFF + 4 * MQ * 4;
    static constexpr int UNIT_OFF = WSUM_OFF + 4 * MQ * 4;  // int[8] ring of unit ids (chunk index), -1 = end
    // Outbox for the partials of small groups (<= OB_GROUP rows): rows wait in LDS and leave in one burst.
    // A store by a compute wave waits ~0.5 us in the CU's vector-memory FIFO behind the loaders' DMA,
    // whatever its size, and the compute waves are the critical path of the barrier-coupled pipeline.
    static constexpr int OB_ROWS = 16;
    static constexpr int OB_GROUP = 4;
    static constexpr int OB_OFF = UNIT_OFF + 32;                 // float [OB_ROWS][D]
    static constexpr int OB_LSE_OFF = OB_OFF + OB_ROWS * D * 4;  // float [OB_ROWS]
    static constexpr int OB_DST_OFF = OB_LSE_OFF + OB_ROWS * 4;  // int32 [OB_ROWS]: partial row index (head-major)
    static constexpr int BYTES = OB_DST_OFF + OB_ROWS * 4;
    static_assert(BYTES <= 160 * 1024, "LDS budget");
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

template <int D>
__global__ __launch_bounds__(512, 2) void stage1_stream_kernel(StreamParams sp) {
    constexpr int KS = D / 16;
    constexpr int LPT = 32 * (D / 8) / 64;  // DMA instructions per loader wave per K (or V) tile: its 32 keys
    static_assert(D == 128, "streaming stage 1 is instantiated for head_dim 128");
    using SM = StreamSmem<D>;
    const Stage1Params& p = sp.s;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* sP = reinterpret_cast<_Float16*>(smem + SM::P_OFF);
    float* sWmax = reinterpret_cast<float*>(smem + SM::WMAX_OFF);
    float* sWsum = reinterpret_cast<float*>(smem + SM::WSUM_OFF);
    int* sUnit = reinterpret_cast<int*>(smem + SM::UNIT_OFF);

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    const int bid = blockIdx.x;
    const int W = (int)gridDim.x;
    const unsigned Hkv = (unsigned)p.Hkv;
    auto head_of = [&](int u) { return (int)((unsigned)u % Hkv); };  // unit id = chunk index = record * Hkv + head
    auto rec_of = [&](int u) { return (int)((unsigned)u / Hkv); };
    const bool is_loader = w >= 4;  // waves 4..7 stream; waves 0..3 compute
    const int lw = w - 4;           // loader lw stages keys [32 lw, 32 lw + 32)

    auto meta = [&](int b) { return smem + SM::META_OFF + b * PLAN_BYTES; };
    auto issue_meta = [&](int rec, int b) {  // loader 0 only: 2 x 1 KB
        const char* src = sp.plan + (int64_t)rec * PLAN_BYTES + 16 * l;
        dma16(src, SM::META_OFF + (uint32_t)b * PLAN_BYTES);
        dma16(src + 1024, SM::META_OFF + (uint32_t)b * PLAN_BYTES + 1024u);
    };

    // Ramp: the plan records of the first three (static) units are fetched BEFORE the record count is known
    // (their addresses depend only on the grid; the plan buffer holds sp.cap records, unwritten ones are never
    // used), so the header read and the record fetch are one round trip instead of two.
    if (w == 4) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int rec = rec_of(j * W + bid);
            if (rec < sp.cap) issue_meta(rec, j);
        }
    }
    if (sp.dbg && tid == 0) {
        sp.dbg[((int64_t)bid * 16 + 15) * 8 + 6] = wall_clock64();
        sp.dbg[((int64_t)bid * 16 + 15) * 8 + 0] = __builtin_amdgcn_s_memtime();
    }
    const int RH = sp.hdr[0];  // records per KV head (written by the plan kernels)
    const int U = RH * p.Hkv;  // units
    // static rounds J (every workgroup owns unit i*W + bid for i < J), then tickets: unit J*W + ticket
    const int per = U / W;
    int J = per - (per * sp.dyn_pct + 99) / 100;
    const bool dynamic = per >= 6 && J >= 3 && sp.dyn_pct > 0;
    if (!dynamic) J = 0x3fffffff;
    auto static_unit = [&](int i) {
        const int u = i * W + bid;
        return (i < J && u < U) ? u : -1;
    };

    // ---- loop-invariant lane constants -------------------------------------------------
    // DMA: instruction i of a tile stages keys 32 lw + 4i + (l>>4), LDS chunk position l&15.
    //   K source chunk = pos ^ (key & 15) = (pos ^ (l>>4)) ^ 4*(i&3);  V source chunk = pos ^ 4*(key&3)
    const int dpos = l & 15, dkey = l >> 4;
    int kchunk_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kchunk_b[j] = ((dpos ^ dkey) ^ (4 * j)) * 16;
    const int vchunk_b = (dpos ^ (4 * (dkey & 3))) * 16;
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

    int64_t rowoff[LPT];  // pool byte offsets of this loader lane's LPT rows of one tile
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
    auto read_desc = [&](int b) {  // {n_vrows, prow, opens_run, run_id}, wave-uniform
        intx4 d = *reinterpret_cast<const intx4*>(meta(b) + PLAN_DESC);
        intx4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(d[j]);
        return r;
    };
    auto stamp = [&](int i, int k) {
        if (sp.dbg && tid == 0 && i < 15) sp.dbg[((int64_t)bid * 16 + i) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    // Q rows of the group that opens at a tile, staged by the loaders in the Q buffer.  Layout like K: row c,
    // 16-byte chunks XOR-ed by (c & 15).  Rows beyond n_vrows alias the first query vector of the group (their
    // mask bits are 0).  2 DMA instructions per loader.
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
    auto finish = [&]() {  // the last workgroup to leave re-arms the scheduler words for the next launch
        if (tid == 0) {
            if (sp.dbg) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                sp.dbg[((int64_t)bid * 16 + 15) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
                sp.dbg[((int64_t)bid * 16 + 15) * 8 + 7] = wall_clock64();
                sp.dbg[((int64_t)bid * 16 + 15) * 8 + 1] = __builtin_amdgcn_s_memtime();
            }
            if (atomicAdd(sp.sched, 1) == W - 1) {
                sp.sched[0] = 0;
                for (int k = 0; k < NTICKET; ++k) sp.sched[ticket_word(k)] = 0;
            }
        }
    };

    // ---- fused paged append: new-token row j is copied into the pool by workgroup W-1 - j % W (nobody
    //      reads those pool rows in this launch: the loaders take them from k_new / v_new) -------------------
    for (int copy_job = W - 1 - bid; copy_job < sp.n_new; copy_job += W) {
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
    if (tid == 0) {
        sUnit[0] = static_unit(0);
        sUnit[1] = static_unit(1);
        sUnit[2] = static_unit(2);
    }
    if (is_loader) wait_vm<0>();  // the three speculative plan records
    lds_barrier();                // sUnit[0..2] and plan records 0..2 visible
    if (static_unit(0) < 0) {
        finish();
        return;
    }

    if (is_loader) {
        // ============================ loader waves ============================
        int u0 = static_unit(0), u1 = static_unit(1), u2 = static_unit(2);
        // Ticket prefetch (loader 0, lane 0).  The unit of stream position q >= J is J*W + ticket; its ticket is
        // requested after barrier H(q-5) (positions 3 and 4: in the prologue), as the LAST operation of that
        // iteration, so the counted wait at A(q-4) may leave it outstanding and the wait at A(q-3) -- it is older
        // than that tile's K/V -- retires it; it is read at C(q-3).  Two tickets are in flight, in the FIXED
        // registers v200 / v201 (by parity of q), named in the asm text and declared clobbered, never bound to a
        // C++ variable: hipcc would otherwise copy the variable (loop phi moves) while the atomic is still in
        // flight and read a stale register -- the hardware has no interlock for that, only s_waitcnt.  The kernel
        // uses ~130 VGPRs, so the allocator never touches v200/v201 (`make asm`, grep v20[01]).
        const bool publisher = lw == 0;
        // NTICKET counters, one cache line each, workgroup b uses counter b % NTICKET (its XCD under round-robin
        // dispatch) and that counter's stripe of the dynamic units: one counter for all 256 workgroups serialises
        // at ~80 atomics/us, which is slower than the tiles are consumed.
        const int tk_lane = bid % NTICKET;
        int* const tk_ptr = sp.sched + ticket_word(tk_lane);
        auto fetch_ticket = [&](int q) {
            if (l == 0) {
                if (q & 1) asm volatile("global_atomic_add v201, %0, %1, off sc0" ::"v"(tk_ptr), "v"(1) : "memory", "v201");
                else asm volatile("global_atomic_add v200, %0, %1, off sc0" ::"v"(tk_ptr), "v"(1) : "memory", "v200");
            }
        };
        auto read_ticket = [&](int q) {
            int t;
            if (q & 1) asm volatile("v_readfirstlane_b32 %0, v201" : "=s"(t)::"memory");
            else asm volatile("v_readfirstlane_b32 %0, v200" : "=s"(t)::"memory");
            return t;
        };
        if (publisher && dynamic) {
            if (J <= 3) fetch_ticket(3);
            if (J <= 4) fetch_ticket(4);
        }
        load_rowoff(0);
        issue_q(0, head_of(u0));  // the first tile of a stream always opens a group
        issue_k(head_of(u0), 0);
        issue_v(head_of(u0), 0);
        if (u1 >= 0) {
            load_rowoff(1);
            issue_k(head_of(u1), 1);
            issue_v(head_of(u1), 1);
        }
        int head_prev = head_of(u0), run_prev = read_desc(0)[3];
        int pos3_done = 0;     // publisher: the stream has ended (no unit(i+3) to look for)
        bool tk_recent = false;  // publisher: a ticket was requested after barrier H(i-1)
        for (int i = 0;; ++i) {
            // A(i): everything of tile i (and the plan records up to i+2, and the ticket of position i+3) has landed
            if (u1 < 0) wait_vm<0>();
            else if (tk_recent) wait_vm<2 * LPT + 1>();
            else wait_vm<2 * LPT>();
            lds_barrier();
            lds_barrier();  // C(i): QK^T(i) done (K stage i&1 free), Q(i) in registers, masks(i) consumed
            if (publisher) {  // unit(i+3): static round or ticket; its plan record
                int u3 = -1;
                if (!pos3_done) {
                    if (i + 3 < J) {
                        u3 = static_unit(i + 3);
                    } else if (dynamic) {
                        u3 = J * W + read_ticket(i + 3) * NTICKET + tk_lane;
                        if (u3 >= U) u3 = -1;
                    }
                    if (u3 < 0) pos3_done = 1;
                }
                if (l == 0) sUnit[(i + 3) & 7] = u3;
                if (u3 >= 0) issue_meta(rec_of(u3), i % 3);
            }
            if (u1 >= 0) {
                const int hn = head_of(u1), rn = read_desc((i + 1) % 3)[3];
                if (hn != head_prev || rn != run_prev) issue_q((i + 1) % 3, hn);  // tile i+1 opens a group
                head_prev = hn;
                run_prev = rn;
            }
            if (u2 >= 0) {
                load_rowoff((i + 2) % 3);
                issue_k(head_of(u2), i & 1);
            }
            lds_barrier();  // F(i)
            lds_barrier();  // H(i): PV(i) done (V stage i&1 free)
            if (u2 >= 0) issue_v(head_of(u2), i & 1);
            if (u1 < 0) break;
            tk_recent = publisher && dynamic && !pos3_done && i + 5 >= J;
            if (tk_recent) fetch_ticket(i + 5);
            u0 = u1;
            u1 = u2;
            u2 = __builtin_amdgcn_readfirstlane(sUnit[(i + 3) & 7]);  // published at C(i)
        }
        return;
    }

    // ============================ compute waves ============================
    half8 qf[KS];
    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    int g_orow = 0;   // this lane's partial row of the open group, relative to kvh * G * rows
    int g_nv = 0;     // virtual rows of the open group
    int ob_used = 0;  // outbox rows in use
    bool qvalid = false;
    int prev_head = -1, prev_run = -1;
    int ucur = static_unit(0);
    int unext = static_unit(1);
    for (int i = 0;; ++i) {
        const int mb = i % 3, mn = (i + 1) % 3;  // plan record slots of this unit and the next
        const int stg = i & 1;                   // K/V stage of this unit
        const bool last = unext < 0;
        const int kvh = head_of(ucur);
        const intx4 cur = read_desc(mb);
        const int orow_c = reinterpret_cast<const int32_t*>(meta(mb) + PLAN_OROW)[c];
        // ticket-assigned tiles (positions >= J) never fold with their neighbours: which tiles meet in one workgroup
        // depends on arrival order there, and the partials -- hence the output bits -- must not
        const bool g_start = (i == 0) || i >= J || kvh != prev_head || cur[3] != prev_run;

        // ---- A: tile i, plan record (i+1) and unit(i+2) are there -------------------------
        stamp(i, 0);
        lds_barrier();
        stamp(i, 1);
        if (g_start) {  // new group: its Q rows were staged by the loaders
            g_orow = orow_c;
            g_nv = cur[0];
            qvalid = c < cur[0];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[ks] = *reinterpret_cast<const half8*>(smem + SM::Q_OFF + c * TILE * 2 + (((2 * ks + h) ^ (c & 15)) * 16));
            m_run = -INFINITY;
            l_run = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
        }
        const bool g_end = last || i + 1 >= J || head_of(unext) != kvh || read_desc(mn)[3] != cur[3];
        const int u2 = __builtin_amdgcn_readfirstlane(sUnit[(i + 2) & 7]);  // published at C(i-1) (or the prologue)

        // ---- B: S^T for this wave's 32 keys, scale, mask, row max --------------------
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!(p.ablate & 1)) {  // profiling knob: 1 = skip QK^T, 2 = skip softmax/P, 4 = skip PV
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 a = *reinterpret_cast<const half8*>(smem + SM::K_OFF + stg * SM::STAGE + krow_b + (kcol_b ^ (32 * ks)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], acc, 0, 0, 0);
            }
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

        // ---- E: online softmax update, P (fp16) -> LDS -----------------------------------
        const float m_tile = fmaxf(fmaxf(sWmax[c], sWmax[MQ + c]), fmaxf(sWmax[2 * MQ + c], sWmax[3 * MQ + c]));
        const float m_new = fmaxf(m_run, m_tile);
        const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
        float sum = 0.f;
        if (!(p.ablate & 2))
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

        // ---- F: V(i) landed; P and row sums visible ----------------------------------------
        stamp(i, 4);
        lds_barrier();
        stamp(i, 5);

        // ---- G: O^T = alpha * O^T + V^T P^T, 32 output columns per wave --------------------
        if (!(p.ablate & 4)) {
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

        // ---- J: bookkeeping for the merge; partial out at the end of a group.  BEFORE barrier H (the loaders
        //      are parked there, their V burst has not been issued yet).  Small groups go to the LDS outbox.
        const int64_t head_rows = (int64_t)kvh * p.G * p.rows;
        if (!(p.ablate & 8) && !g_start && qvalid && w == 0 && h == 0)
            p.partial_lse[head_rows + orow_c] = -INFINITY;  // row folded into its group's partial
        const bool to_outbox = g_end && g_nv <= SM::OB_GROUP && !(p.ablate & 64);  // wave-uniform
        if (to_outbox) {
            if (qvalid) {
                const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
                float* ob = reinterpret_cast<float*>(smem + SM::OB_OFF) + (ob_used + c) * D + 32 * w + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    floatx4 v4 = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
                    *reinterpret_cast<floatx4*>(ob + 8 * g4) = v4;
                }
                if (w == 0 && h == 0) {
                    reinterpret_cast<float*>(smem + SM::OB_LSE_OFF)[ob_used + c] =
                        (l_run > 0.f) ? (m_run + __builtin_amdgcn_logf(l_run)) * LN2 : -INFINITY;
                    reinterpret_cast<int32_t*>(smem + SM::OB_DST_OFF)[ob_used + c] = (int32_t)(head_rows + g_orow);
                }
            }
            ob_used += g_nv;
        }
        if (!(p.ablate & 16) && g_end && qvalid && !to_outbox) {
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

        // ---- H: every wave is done with sV and sP -------------------------
        stamp(i, 6);
        lds_barrier();
        stamp(i, 7);
        // outbox flush: when another small group might not fit, and at the end of the stream (the loaders are
        // done then and the FIFO is empty).  Reads here, next outbox writes after barriers A..F of tile i+1.
        if (ob_used > SM::OB_ROWS - SM::OB_GROUP || (last && ob_used > 0)) {
            if (!(p.ablate & 16)) {
                const int32_t* dst = reinterpret_cast<const int32_t*>(smem + SM::OB_DST_OFF);
                for (int r = w; r < ob_used; r += 4) {
                    const floatx2 v2 = *reinterpret_cast<const floatx2*>(smem + SM::OB_OFF + r * D * 4 + l * 8);
                    *reinterpret_cast<floatx2*>(p.partial_o + (int64_t)dst[r] * D + 2 * l) = v2;
                }
                if (w == 0 && l < ob_used) p.partial_lse[dst[l]] = reinterpret_cast<const float*>(smem + SM::OB_LSE_OFF)[l];
            }
            ob_used = 0;
        }
        if (last) break;
        prev_head = kvh;
        prev_run = cur[3];
        ucur = unext;
        unext = u2;
    }
    finish();
}

}  // namespace deft
