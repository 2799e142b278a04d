// Stage 1 (Flatten and Node), streaming form: persistent workgroups, LDS-DMA pipeline.
//
// Included by deft_kernels.hip (needs its typedefs and Stage1Params).
//
//   * plan kernels (once per decode step) pack each 128-slot KV tile's metadata into one 2 KB record:
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
    int32_t* flags; // bit 0: opens a run (query list differs from the previous unit's); bits 1..: unit index of the run's first unit
    int32_t* prow;  // first partial row of the unit's tile
    // tile-parallel record order (stage1_np.h), indexed by RECORD: which unit the record packs, and its chunk
    int32_t* perm;   // record -> unit
    int32_t* ch_n;   // tiles of the chunk this record leads (itself included); 0 = follower
    int32_t* ch_fb;  // record index of the chunk's first follower (followers are consecutive records)
    // union groups (Flatten, tile-parallel order), indexed by GROUP id = aux - 1: consecutive leaf tiles whose query
    // lists differ but are small are folded by ONE workgroup over the union of their queries
    int32_t* gn;    // queries in the union (<= UNION_CAP)
    int32_t* gq;    // [UNION_CAP][cap] query rows of the union, ascending
    int32_t* grow;  // [UNION_CAP][cap] the partial row that carries each union query (its first occurrence in the group)
};
constexpr int UNION_CAP = 4;

// Record order for the tile-parallel stage 1 (one workgroup per chunk, stage1_np.h).  A run of `nt` units with
// one query list is cut into S = ceil(nt / C) chunks; chunk p folds units p, p + S, p + 2S, ... of the run
// (interleaved: the chunks of a run advance through the pool side by side, one contiguous front).  Records are
// renumbered: the leaders of all chunks first (run by run, so the long shared-prefix chunks are dispatched first),
// then the followers, those of one chunk consecutive.  hdr[1] = number of leaders.  One thread.
// The runs come from a table the emitting thread kept in LDS (run k = units r0[k] .. r0[k] + nt[k] - 1, uni[k] = it is
// a union group): walking the unit arrays in global memory instead costs one dependent load per unit (~100 us for the
// north-star tree, once per decode step).
struct RunTable {
    int* r0;
    int* nt;
    int* uni;
    int n;    // runs recorded
    int cap;  // capacity; n > cap = overflow, fall back to scanning the unit arrays
};
__device__ inline void run_push(RunTable& rt, int r0, int nt, int uni) {
    if (rt.n < rt.cap) {
        rt.r0[rt.n] = r0;
        rt.nt[rt.n] = nt;
        rt.uni[rt.n] = uni;
    }
    ++rt.n;
}

__device__ inline void np_record_order(const UnitList& ul, int R, int Hkv, int G, int slots, int chunk_c, int32_t* hdr,
                                       RunTable rt) {
    if (rt.n > rt.cap) {  // rebuild the table is impossible: scan (slow path, huge trees only)
        rt.n = 0;
        rt.cap = 0;
    }
    const bool have = rt.cap > 0;
    // run iteration: either the LDS table or a scan over flags / aux
    auto for_runs = [&](auto&& fn) {
        if (have) {
            for (int k = 0; k < rt.n; ++k) fn(rt.r0[k], rt.nt[k], rt.uni[k]);
        } else {
            for (int r = 0; r < R;) {
                const int id = ul.flags[r] >> 1;
                int e = r + 1;
                while (e < R && (ul.flags[e] >> 1) == id) ++e;
                fn(r, e - r, ul.aux[r] > 0 ? 1 : 0);
                r = e;
            }
        }
    };
    int C = chunk_c;
    if (C <= 0) {
        // Measured on MI355X (tools/np_sweep.sh): per-workgroup cost (descriptor round trip, 32-row epilogue) favours
        // long chunks, the critical path and the number of resident slots bound them.  8 tiles for long shared
        // prefixes, 4 from 8 tiles on, halved while fewer than ~0.3 workgroups per slot would be left.
        int lmax = 0;
        for_runs([&](int, int nt, int uni) {
            if (!uni && nt > lmax) lmax = nt;
        });
        C = lmax >= 32 ? 8 : (lmax >= 8 ? 4 : (lmax >= 4 ? 2 : 1));
        // ... and with GQA a chunk should not outlast the launch: the passes of a shared tile are separate chunks that
        // hit L2, so with T tiles per resident slot in all, chunks longer than ~T leave a few workgroups running
        // alone at the end (Llama-3 north-star tree: 2.8 tiles per slot, 8-tile chunks ran 18 us of a 25 us launch
        // with a quarter of the slots occupied; 4-tile chunks: 19.9 us).  MHA, every tile from HBM, measured the
        // other way (1-token branches, 2 tiles per slot: 8-tile chunks 19.1 us, 4-tile chunks 21.0).
        if (G > 1) {
            int64_t tiles_all = 0;
            for_runs([&](int, int nt, int) { tiles_all += nt; });
            int cmax = 1;
            while (cmax < 8 && (int64_t)cmax * slots < tiles_all * Hkv) cmax <<= 1;
            if (C > cmax) C = cmax;
        }
        for (; C > 1; C >>= 1) {
            int64_t n = 0;
            for_runs([&](int, int nt, int uni) { n += uni ? 1 : (nt + C - 1) / C; });
            if (10 * n * Hkv >= 3LL * slots) break;
        }
    }
    int NL = 0;
    for_runs([&](int, int nt, int uni) { NL += uni ? 1 : (nt + C - 1) / C; });
    int li = 0, fi = NL;
    for_runs([&](int r, int nt, int uni) {
        const int S = uni ? 1 : (nt + C - 1) / C;  // a union group is one chunk
        for (int p = 0; p < S; ++p) {
            const int cnt = (nt - p + S - 1) / S;
            ul.perm[li] = r + p;
            ul.ch_n[li] = cnt;
            ul.ch_fb[li] = fi;
            ++li;
            for (int j = 1; j < cnt; ++j, ++fi) {
                ul.perm[fi] = r + p + j * S;
                ul.ch_n[fi] = 0;
                ul.ch_fb[fi] = 0;
            }
        }
    });
    hdr[1] = NL;
}

// Record order of the tile-parallel stage 1, written by all waves of the unit kernel from its LDS run table (the
// rules of np_record_order above, same result).  Called by every thread after the units are written; rT0 / rSp are
// scratch arrays of run_cap words (the callers' run fields are dead by then), sMeta[2..3] two shared words.
__device__ inline void record_order_parallel(const UnitList& ul, const RunTable& rt, int NR, int* rT0, int* rSp, int* sMeta,
                                             int32_t* hdr, int Hkv, int G, int slots, int chunk_c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    __syncthreads();  // (rT0 / rSp are reused below)
    // Wave 0: chunk length C from sums / maxima over the runs, then each run's first leader and first follower record
    // by prefix sums over the runs' chunk counts.
    if (threadIdx.x < 64) {
        auto wave_sum = [&](auto&& f) {
            int acc = 0;
            for (int k = lane; k < NR; k += 64) acc += f(rt.nt[k], rt.uni[k]);
            for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
            return acc;
        };
        int C = chunk_c;
        if (C <= 0) {
            int lmax = 0;
            for (int k = lane; k < NR; k += 64)
                if (!rt.uni[k] && rt.nt[k] > lmax) lmax = rt.nt[k];
            for (int m = 32; m > 0; m >>= 1) lmax = max(lmax, __shfl_xor(lmax, m, 64));
            C = lmax >= 32 ? 8 : (lmax >= 8 ? 4 : (lmax >= 4 ? 2 : 1));
            if (G > 1) {
                const int64_t tiles_all = wave_sum([](int nt, int) { return nt; });
                int cmax = 1;
                while (cmax < 8 && (int64_t)cmax * slots < tiles_all * Hkv) cmax <<= 1;
                if (C > cmax) C = cmax;
            }
            for (; C > 1; C >>= 1) {
                const int64_t n = wave_sum([C](int nt, int uni) { return uni ? 1 : (nt + C - 1) / C; });
                if (10 * n * Hkv >= 3LL * slots) break;
            }
        }
        int lead = 0, foll = 0;
        for (int base = 0; base < NR; base += 64) {
            const int k = base + lane;
            const int nt = k < NR ? rt.nt[k] : 0;
            const int S = k < NR ? (rt.uni[k] ? 1 : (nt + C - 1) / C) : 0;
            int a = S, b = nt - S;  // inclusive scans over the lanes
            for (int d = 1; d < 64; d <<= 1) {
                const int ua = __shfl_up(a, d, 64), ub = __shfl_up(b, d, 64);
                if (lane >= d) {
                    a += ua;
                    b += ub;
                }
            }
            if (k < NR) {
                rT0[k] = lead + a - S;
                rSp[k] = foll + b - (nt - S);
            }
            lead += __shfl(a, 63, 64);
            foll += __shfl(b, 63, 64);
        }
        if (lane == 0) {
            sMeta[2] = C;
            sMeta[3] = lead;
            hdr[1] = lead;
        }
    }
    __syncthreads();
    {
        const int C = sMeta[2], NL = sMeta[3];
        for (int k = wave; k < NR; k += nwaves) {
            const int first = rt.r0[k], nt = rt.nt[k];
            const int S = rt.uni[k] ? 1 : (nt + C - 1) / C;  // a union group is one chunk
            const int li = rT0[k], fi = NL + rSp[k];
            const int q = nt / S, rem = nt - q * S;  // chunk p folds units p, p + S, ...: q + 1 of them for p < rem, else q
            for (int u = lane; u < nt; u += 64) {
                const int j = u / S, pc = u - j * S;
                const int fb = fi + pc * (q - 1) + min(pc, rem);  // followers of the chunks before pc
                if (j == 0) {
                    ul.perm[li + pc] = first + pc;
                    ul.ch_n[li + pc] = q + (pc < rem ? 1 : 0);
                    ul.ch_fb[li + pc] = fb;
                } else {
                    ul.perm[fb + j - 1] = first + u;
                    ul.ch_n[fb + j - 1] = 0;
                    ul.ch_fb[fb + j - 1] = 0;
                }
            }
        }
    }
}

// Union group of leaf tiles starting at block t (Flatten, tile-parallel order): up to `ulen` consecutive blocks whose
// query lists hold at most `ucap` queries each and at most `ucap` distinct queries together.  Returns the number of
// blocks taken; uq / urow / un = the union's queries in order of first occurrence and the partial row (block_q
// position) of each first occurrence.  sQ: the [NB][UNION_CAP] LDS table of the small blocks' lists, or nullptr
// (lists read from block_q).  Everything lives in registers, every loop is unrolled over UNION_CAP.
__device__ inline int union_group(int t, int NB, int ulen, int ucap, const int* sCnt, const int* sOff, const int* sQ,
                                  const int64_t* block_q, int (&uq)[UNION_CAP], int (&urow)[UNION_CAP], int& un) {
    static_assert(UNION_CAP == 4, "the query table is read as int4");
    un = 0;
#pragma unroll
    for (int j = 0; j < UNION_CAP; ++j) uq[j] = 0, urow[j] = 0;
    int te = t;
    while (te < NB && te - t < ulen) {
        const int cnt = sCnt[te];
        if (cnt > ucap) break;
        const int off = sOff[te];
        int qv[UNION_CAP];
        if (sQ) {
            const int4 v = *reinterpret_cast<const int4*>(sQ + te * UNION_CAP);
            qv[0] = v.x, qv[1] = v.y, qv[2] = v.z, qv[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < UNION_CAP; ++i) qv[i] = i < cnt ? (int)block_q[off + i] : 0;
        }
        int add = 0;  // queries of block te that are new to the union
#pragma unroll
        for (int i = 0; i < UNION_CAP; ++i) {
            bool found = false;
#pragma unroll
            for (int j = 0; j < UNION_CAP; ++j) found |= (j < un) & (uq[j] == qv[i]);
            add += (i < cnt && !found) ? 1 : 0;
        }
        if (un + add > ucap) break;
#pragma unroll
        for (int i = 0; i < UNION_CAP; ++i) {
            bool found = i >= cnt;
#pragma unroll
            for (int j = 0; j < UNION_CAP; ++j) found |= (j < un) & (uq[j] == qv[i]);
            if (!found) {  // first occurrence: this tile's row carries the query's partial
#pragma unroll
                for (int j = 0; j < UNION_CAP; ++j)
                    if (j == un) {
                        uq[j] = qv[i];
                        urow[j] = off + i;
                    }
                ++un;
            }
        }
        ++te;
    }
    return te - t;
}

// Flatten: one workgroup.  Phase 1 (parallel over blocks): does block t open a run, how many passes.
// Phase 2 (one thread): emit units run by run, pass-major inside a run so that consecutive units fold.
// Tile-parallel order only (union_len > 1): short runs of leaf tiles with small, different query lists -- a branch's
// tail shares a block with the next branch's head, so their lists go {a}, {a,b}, {b}, {b,c} ... -- are grouped, up
// to union_len tiles and UNION_CAP queries, into ONE run over the union of their queries (per-slot masks keep a
// query away from keys that are not on its path), so that one workgroup folds them and writes one partial per query.
__global__ __launch_bounds__(1024) void flatten_units_kernel(const int64_t* block_q, const int64_t* block_q_cnts,
                                                            const int64_t* block_q_offset, int NB, int G, int cap,
                                                            UnitList ul, int32_t* hdr, int32_t* sched, int np, int Hkv,
                                                            int slots, int chunk_c, int union_len, int taper, int run_cap,
                                                            int qtab, int par) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sOpen = reinterpret_cast<int*>(smem);  // [NB]
    int* sPass = sOpen + NB;                    // [NB]
    int* sCnt = sPass + NB;                     // [NB] block_q_cnts
    int* sOff = sCnt + NB;                      // [NB] block_q_offset
    // [NB][UNION_CAP] query lists of the blocks small enough to join a union group (qtab: the table fits in LDS):
    // the one-thread phase below otherwise waits for a global load per leaf tile
    int* sQ = sOff + NB;
    int* sRun = sQ + (qtab ? UNION_CAP * NB : 0);
    RunTable rt{sRun, sRun + run_cap, sRun + 2 * run_cap, 0, run_cap};
    for (int t = threadIdx.x; t < NB; t += blockDim.x) {
        const int cnt = (int)block_q_cnts[t];
        sCnt[t] = cnt;
        sOff[t] = (int)block_q_offset[t];
        sPass[t] = (cnt * G + MQ - 1) / MQ;
    }
    __syncthreads();
    // Does block t's query list differ from the one 1 / 2 / 3 / 4 blocks back?  Half a wave per block, lane i on
    // list entry i (and i + 32, ... for longer lists), the five loads of an entry independent of each other: one
    // round trip per block instead of one per list entry (a shared-prefix block has 32 entries, and a thread walking
    // them with early exit waited ~1 us for each).
    //   bit 0: opens a run;  bits 1..3: the list also differs from the one 2 / 3 / 4 blocks back (a node with more
    //   than 32 queries is emitted by the reference as alternating blocks -- queries 0..31 / 32..63 / ... of the same
    //   128 slots, tree_cache.py:763-799 -- so its blocks repeat with period ceil(queries / 32)); inside an ordinary
    //   run the longer periods are never looked at and read as "differs".
    {
        const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
        const int nhw = (blockDim.x >> 6) * 2;
        for (int t0 = (threadIdx.x >> 6) * 2; t0 < NB; t0 += nhw) {
            const int t = t0 + half;
            const bool live = t < NB;
            const int cnt = live ? sCnt[t] : 0;
            const int a = live ? sOff[t] : 0;
            int diff = 0;  // bit pd-1: some entry of this lane differs from the block pd back
            bool same_cnt[4];
            for (int pd = 1; pd <= 4; ++pd) same_cnt[pd - 1] = live && t >= pd && sCnt[t - pd] == cnt;
            for (int i = li; i < cnt; i += 32) {
                const int64_t mine = block_q[a + i];
                int64_t other[4];
                for (int pd = 1; pd <= 4; ++pd) other[pd - 1] = same_cnt[pd - 1] ? block_q[sOff[t - pd] + i] : mine;
                for (int pd = 1; pd <= 4; ++pd) diff |= (other[pd - 1] != mine) ? (1 << (pd - 1)) : 0;
                if (qtab && cnt <= UNION_CAP) sQ[t * UNION_CAP + i] = (int)mine;
            }
            int bits = 0;
            for (int pd = 1; pd <= 4; ++pd) {
                const unsigned long long b = __ballot((diff >> (pd - 1)) & 1);
                const bool any = ((half ? (b >> 32) : b) & 0xffffffffull) != 0;
                bits |= (!same_cnt[pd - 1] || any) ? (1 << (pd - 1)) : 0;
            }
            if (!(bits & 1)) bits = 0xe;  // not opening a run: longer periods read as "differs"
            if (live && li == 0) sOpen[t] = bits;
        }
    }
    __syncthreads();
    // Phase 1b (tile-parallel order): for every block, how many blocks a union group starting there would take
    // (0 = none), one thread per block, into bits 8.. of sOpen -- the walk below then only looks the answer up.
    const int ucap = (UNION_CAP * G <= MQ) ? UNION_CAP : MQ / G;  // union queries whose virtual rows fit one pass
    auto union_len_at = [&](int t) {
        // taper: the workgroups dispatched last should be short, so that the launch ends together -- the last
        // `slots` leaf tiles (per head) stay single, the `2 slots` before them go in pairs
        int ulen = union_len;
        if (ulen <= 0) ulen = G > 1 ? 1 : ((int64_t)NB * Hkv < 2048 ? 4 : 3);  // measured, tools/np_sweep.sh / tools/ab.py
        if (taper) {
            const int64_t rest = (int64_t)(NB - t) * Hkv;
            if (rest <= (int64_t)slots) ulen = 1;
            else if (rest <= 3LL * slots) ulen = min(ulen, 2);
        }
        return ulen;
    };
    if (np && ucap >= 2)
        for (int t = threadIdx.x; t < NB; t += blockDim.x) {
            const int ulen = union_len_at(t);
            int g = 0;
            if (ulen > 1 && sCnt[t] <= ucap) {
                bool short_run = NB - t < ulen;  // the run that opens at t is shorter than a group
                for (int u = t + 1; u < t + ulen && u < NB; ++u) short_run |= (sOpen[u] & 1) != 0;
                if (short_run) {
                    int uq[UNION_CAP], urow[UNION_CAP], un;
                    g = union_group(t, NB, ulen, ucap, sCnt, sOff, qtab ? sQ : nullptr, block_q, uq, urow, un);
                    if (g < 2) g = 0;
                }
            }
            sOpen[t] = (sOpen[t] & 0xf) | (g << 8);
        }
    __syncthreads();
    // Phase 2: wave 0 walks the blocks and decides the runs (every lane takes the same decisions; the lanes only split
    // the searches for the next run boundary).  `par`: every run gets an entry of the LDS table, and the units and the
    // record order are then written by all waves (phases 3 and 4) -- one thread emitting them costs ~0.4 us per unit
    // and pass, 75 us per decode step for the north-star tree and 2 ms for a 100k-token prefix under 48 branches.
    // Otherwise (tables beyond the LDS) lane 0 emits as it walks.
    int* sMeta = sRun + (par ? 5 : 3) * run_cap;  // [8]: units, runs, chunk length, leaders, "written by all waves"
    int* rT0 = sRun + 3 * run_cap;    // par: first block of the run;      later: the run's first leader record
    int* rSp = sRun + 4 * run_cap;    // par: block stride | pass << 8;   later: the run's first follower record - leaders
    const int lane = threadIdx.x & 63;
    const int par_req = par;
    if (threadIdx.x < 64) {
      // (a second walk, lane 0 emitting, if the runs did not fit the table: trees with very many small runs)
      for (int attempt = 0; attempt < 2; ++attempt) {
        par = par_req && attempt == 0;
        rt.n = 0;
        int r = 0, ng = 0;
        // first block >= from whose sOpen has a bit of `mask` set (NB if none)
        auto find = [&](int from, int mask) {
            for (int base = from; base < NB; base += 64) {
                const int t = base + lane;
                const unsigned long long b = __ballot(t < NB && (sOpen[t] & mask));
                if (b) return base + __ffsll((long long)b) - 1;
            }
            return NB;
        };
        // units of one run: blocks t0, t0 + st, ... < te, pass ps, aux (0, or union group + 1)
        auto emit_run = [&](int t0, int te, int st, int aux, int ps) {
            int n = st == 1 ? te - t0 : (te - t0 + st - 1) / st;
            if (n > cap - r) n = cap - r;
            if (n <= 0) return;
            const int first = r;
            if (par) {
                if (lane == 0 && rt.n < rt.cap) {
                    rt.r0[rt.n] = first;
                    rt.nt[rt.n] = n;
                    rt.uni[rt.n] = aux;
                    rT0[rt.n] = t0;
                    rSp[rt.n] = st | (ps << 8);
                }
                ++rt.n;
            } else {
                if (lane == 0)
                    for (int j = 0; j < n; ++j) {
                        const int t = t0 + j * st;
                        ul.src[first + j] = t;
                        ul.aux[first + j] = aux;
                        ul.pass[first + j] = ps;
                        ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
                        ul.prow[first + j] = sOff[t];
                    }
                if (rt.n < rt.cap && lane == 0) {
                    rt.r0[rt.n] = first;
                    rt.nt[rt.n] = n;
                    rt.uni[rt.n] = aux;
                }
                ++rt.n;
            }
            r += n;
        };
        for (int ta = 0; ta < NB;) {
            // ---- union group starting at ta (phase 1b) ------------------------------------------------------
            const int glen = sOpen[ta] >> 8;
            if (glen >= 2 && r < cap) {
                if (!par && lane == 0) {  // (the parallel form writes the group in phase 3)
                    int uq[UNION_CAP], urow[UNION_CAP], un;
                    union_group(ta, NB, glen, ucap, sCnt, sOff, qtab ? sQ : nullptr, block_q, uq, urow, un);
                    ul.gn[ng] = un;
                    for (int j = 0; j < UNION_CAP; ++j) {
                        ul.gq[j * cap + ng] = j < un ? uq[j] : 0;
                        ul.grow[j * cap + ng] = j < un ? urow[j] : 0;
                    }
                }
                emit_run(ta, ta + glen, 1, ng + 1, 0);
                ++ng;
                ta += glen;
                continue;
            }
            const int tb = find(ta + 1, 1);
            const int passes = sPass[ta];
            const int cnt_a = sCnt[ta];
            // ---- P query chunks of one node, alternating block by block: P interleaved runs ------------------
            if (tb - ta == 1 && cnt_a == MQ) {  // (the first chunk of such a node is always full: cheap filter)
                int P = 0;
                for (int pd = 2; pd <= 4 && !P; ++pd) {
                    bool rep = ta + 2 * pd <= NB;
                    for (int t = ta + pd; rep && t < ta + 2 * pd; ++t) rep = !(sOpen[t] & (1 << (pd - 1)));
                    if (rep) P = pd;
                }
                if (P) {
                    const int te = find(ta + P, 1 << (P - 1));
                    for (int par_ = 0; par_ < P; ++par_) {
                        const int pp = sPass[ta + par_];
                        for (int ps = 0; ps < pp; ++ps) emit_run(ta + par_, te, P, 0, ps);
                    }
                    ta = te;
                    continue;
                }
            }
            for (int ps = 0; ps < passes; ++ps) emit_run(ta, tb, 1, 0, ps);
            ta = tb;
        }
        if (lane == 0) {
            hdr[0] = r;
            hdr[1] = 0;
            sched[0] = 0;
            for (int k = 0; k < NTICKET; ++k) sched[ticket_word(k)] = 0;
            if (par && rt.n <= rt.cap) {
                sMeta[0] = r;
                sMeta[1] = rt.n;
                sMeta[4] = 1;
            } else if (!par) {
                sMeta[4] = 0;
                if (np) np_record_order(ul, r, Hkv, G, slots, chunk_c, hdr, rt);
            }
        }
        if (!par || rt.n <= rt.cap) break;
      }
    }
    __syncthreads();
    if (!sMeta[4]) return;
    // Phase 3: the units of run k, one wave per run, one lane per unit.
    const int NR = sMeta[1];
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int k = wave; k < NR; k += nwaves) {
        const int first = rt.r0[k], n = rt.nt[k], aux = rt.uni[k], t0 = rT0[k], st = rSp[k] & 0xff, ps = rSp[k] >> 8;
        if (aux > 0 && lane == 0) {  // a union group: its queries and the rows that carry their partials
            int uq[UNION_CAP], urow[UNION_CAP], un;
            union_group(t0, NB, sOpen[t0] >> 8, ucap, sCnt, sOff, qtab ? sQ : nullptr, block_q, uq, urow, un);
            ul.gn[aux - 1] = un;
            for (int j = 0; j < UNION_CAP; ++j) {
                ul.gq[j * cap + aux - 1] = j < un ? uq[j] : 0;
                ul.grow[j * cap + aux - 1] = j < un ? urow[j] : 0;
            }
        }
        for (int j = lane; j < n; j += 64) {
            const int t = t0 + j * st;
            ul.src[first + j] = t;
            ul.aux[first + j] = aux;
            ul.pass[first + j] = ps;
            ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
            ul.prow[first + j] = sOff[t];
        }
    }
    if (!np) return;
    // Phase 4: record order of the tile-parallel stage 1
    record_order_parallel(ul, rt, NR, rT0, rSp, sMeta, hdr, Hkv, G, slots, chunk_c);
}

// One workgroup of 128 threads per unit (+ the sentinel): pack its record.
__global__ __launch_bounds__(128) void flatten_records_kernel(const int64_t* block_q, const int64_t* block_q_cnts,
                                                              const int64_t* block_bitmasks, const int64_t* block_kv,
                                                              const int64_t* block_lens, int G, int rows, int64_t q_st,
                                                              int64_t q_sh, int64_t kv_stride_slot, UnitList ul,
                                                              const int32_t* hdr, char* plan, int32_t* row_q,
                                                              const int32_t* cache_loc, int n_new, int64_t new_row_bytes,
                                                              int np) {
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    const int R = hdr[0];
    char* rec = plan + (int64_t)r * PLAN_BYTES;
    int64_t* ro = reinterpret_cast<int64_t*>(rec + PLAN_ROWOFF);
    uint32_t* mk = reinterpret_cast<uint32_t*>(rec + PLAN_MASK);
    int32_t* desc = reinterpret_cast<int32_t*>(rec + PLAN_DESC);
    if (r > R) {  // unused capacity: a tile-parallel workgroup that lands here must see "not a chunk leader"
        if (k == 0) desc[4] = 0;
        return;
    }
    if (r == R) {  // sentinel
        ro[k] = 0;
        mk[k] = 0u;
        if (k == 0) {
            desc[0] = 0;
            desc[1] = 0;
            desc[2] = 1;
            desc[3] = -1;
            desc[4] = 0;
        }
        return;
    }
    const int u = np ? ul.perm[r] : r;  // unit packed into this record
    const int t = ul.src[u];
    const int ps = ul.pass[u];
    const int prow = ul.prow[u];
    const int len = (int)block_lens[t];
    const int cnt = (int)block_q_cnts[t];
    const bool live = k < len;
    ro[k] = plan_rowoff(block_kv[(int64_t)t * TILE + (live ? k : 0)], kv_stride_slot, cache_loc, n_new, new_row_bytes);
    const uint32_t qmask = live ? (uint32_t)block_bitmasks[(int64_t)t * TILE + k] : 0u;
    if (ul.aux[u] > 0) {
        // ---- member of a union group: virtual row v = (union query v / G, head v % G); a query that is not in
        //      this block's own list sees none of its slots ------------------------------------------------
        const int gid = ul.aux[u] - 1;
        const int cap = (int)(ul.gq - ul.gn);  // arrays are `cap` apart
        const int un = ul.gn[gid];
        const int nvu = un * G;
        uint32_t vmu = 0u;
        for (int j = 0; j < un; ++j) {
            const int qv = ul.gq[j * cap + gid];
            int idx = -1;
            for (int i = 0; i < cnt; ++i)
                if ((int)block_q[prow + i] == qv) idx = i;
            if (idx >= 0 && ((qmask >> idx) & 1u)) vmu |= ((G >= 32 ? 0xffffffffu : ((1u << G) - 1u)) << (j * G));
        }
        mk[k] = vmu;
        if (k < MQ) {
            const int kk = k < nvu ? k : 0;  // rows beyond the union alias its first row (their mask bits are 0)
            const int j = kk / G, g = kk % G;
            reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = (int)(ul.gq[j * cap + gid] * q_st + g * q_sh);
            reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = k < nvu ? g * rows + ul.grow[j * cap + gid] : 0;
            if (k < cnt) {  // this tile's own partial rows: live iff the group parks a query's partial there
                int qlive = -1;
                for (int j2 = 0; j2 < un; ++j2)
                    if (ul.grow[j2 * cap + gid] == prow + k) qlive = ul.gq[j2 * cap + gid];
                row_q[prow + k] = qlive;
            }
        }
        if (k == 0) {
            desc[0] = nvu;
            desc[1] = prow;
            desc[2] = ul.flags[u] & 1;
            desc[3] = ul.flags[u] >> 1;
            desc[4] = ul.ch_n[r];
            desc[5] = ul.ch_fb[r];
        }
        return;
    }
    const int nv = min(MQ, cnt * G - MQ * ps);  // virtual rows of this pass
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
        if (!np) {
            if (ps == 0 && k < cnt) row_q[prow + k] = (int32_t)block_q[prow + k];
        } else if (k < nv && (MQ * ps + k) % G == 0) {
            // tile-parallel order: folding is decided here, so only a chunk leader's rows are live
            const int qi = (MQ * ps + k) / G;
            row_q[prow + qi] = ul.ch_n[r] > 0 ? (int32_t)block_q[prow + qi] : -1;
        }
    }
    if (k == 0) {
        desc[0] = nv;
        desc[1] = prow;
        desc[2] = ul.flags[u] & 1;
        desc[3] = ul.flags[u] >> 1;  // run id: tiles with equal ids share one query list and may fold
        desc[4] = np ? ul.ch_n[r] : 0;
        desc[5] = np ? ul.ch_fb[r] : 0;
    }
}

// Node mode (tree_attention.py:14-293): every entry (a node's KV x up to 32 of its queries) is cut into
// 128-slot tiles; all live slots are visible to all of the entry's queries.  Consecutive tiles of one
// entry fold, which is the reference's serial online-softmax walk (:230-276) without the serialisation.
// Small entries (one tile, one pass) that follow each other are PACKED into one tile as long as their slots fit
// in 128 and their virtual query rows in 32 -- per-slot row masks make that the same arithmetic (a Medusa step
// has 64 one-token nodes: 2 tiles instead of 64).  A packed unit has aux = -(entries in it).
// One workgroup.  Wave 0 walks the entries and decides the runs (packs are sequential by nature), all waves then write the units and the record order from the LDS run table -- as in
// flatten_units_kernel; `par` = 0 (tables beyond the LDS) or an overflowing table: lane 0 emits as it walks.
__global__ __launch_bounds__(1024) void node_units_kernel(const int64_t* node_kv_len, const int64_t* node_q_len, int NE, int G,
                                                          int cap, int64_t rows_cap, UnitList ul, int32_t* hdr,
                                                          int32_t* sched, int32_t* row_q, int np, int Hkv, int slots,
                                                          int chunk_c, int run_cap, int par) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sRun = reinterpret_cast<int*>(smem);
    RunTable rt{sRun, sRun + run_cap, sRun + 2 * run_cap, 0, run_cap};
    int* rT0 = sRun + 3 * run_cap;    // par: the run's entry;               later: its first leader record
    int* rSp = sRun + 4 * run_cap;    // par: the run's pass;                later: its first follower record - leaders
    int* rProw = sRun + 5 * run_cap;  // par: partial row of the run's first tile
    int* rQl = sRun + 6 * run_cap;    // par: partial rows per tile
    int* rAux = sRun + 7 * run_cap;   // par: 1 = tiles of one entry (aux = tile index), <= 0 = a pack (aux = -entries)
    int* sMeta = sRun + (par ? 8 : 3) * run_cap;  // [8]
    for (int64_t i = threadIdx.x; i < rows_cap; i += blockDim.x) row_q[i] = -1;
    const int lane = threadIdx.x & 63;
    const int par_req = par;
    if (threadIdx.x < 64) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            par = par_req && attempt == 0;
            rt.n = 0;
            int r = 0, rowbase = 0;
            int pack_r = -1, pack_k = -1, pack_n = 0, pack_keys = 0, pack_rows = 0;  // open pack: unit, run, entries, slots, virtual rows
            auto emit_run = [&](int e, int n, int ps, int prow0, int ql, int aux) {
                if (n > cap - r) n = cap - r;
                if (n <= 0) return;
                const int first = r;
                if (par) {
                    if (lane == 0 && rt.n < rt.cap) {
                        rt.r0[rt.n] = first;
                        rt.nt[rt.n] = n;
                        rt.uni[rt.n] = 0;
                        rT0[rt.n] = e;
                        rSp[rt.n] = ps;
                        rProw[rt.n] = prow0;
                        rQl[rt.n] = ql;
                        rAux[rt.n] = aux;
                    }
                } else if (lane == 0) {
                    for (int j = 0; j < n; ++j) {
                        ul.src[first + j] = e;
                        ul.aux[first + j] = aux > 0 ? j : aux;
                        ul.pass[first + j] = ps;
                        ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
                        ul.prow[first + j] = prow0 + j * ql;
                    }
                    if (rt.n < rt.cap) {
                        rt.r0[rt.n] = first;
                        rt.nt[rt.n] = n;
                        rt.uni[rt.n] = 0;
                    }
                }
                ++rt.n;
                r += n;
            };
            // 64 entries at a time: lane i loads the lengths of entry base + i (one round trip per batch), the walk
            // reads them with v_readlane -- scalar code, no memory access per entry (a Medusa step has 65 entries)
            for (int base = 0; base < NE; base += 64) {
              const int mine = base + lane;
              const int vlen = mine < NE ? (int)node_kv_len[mine] : 0;
              const int vql = mine < NE ? (int)node_q_len[mine] : 0;
              const int lim = min(64, NE - base);
              for (int i = 0; i < lim; ++i) {
                const int e = base + i;
                const int len = __builtin_amdgcn_readlane(vlen, i);
                const int nt = (len + TILE - 1) / TILE;
                const int ql = __builtin_amdgcn_readlane(vql, i);
                const int npass = (ql * G + MQ - 1) / MQ;
                if (nt == 1 && npass == 1 && r < cap) {
                    if (pack_r >= 0 && pack_n < MQ && pack_keys + len <= TILE && pack_rows + ql * G <= MQ) {
                        ++pack_n;
                        pack_keys += len;
                        pack_rows += ql * G;
                        if (lane == 0) {
                            if (!par) ul.aux[pack_r] = -pack_n;
                            else if (pack_k < rt.cap) rAux[pack_k] = -pack_n;
                        }
                    } else {
                        pack_r = r;
                        pack_k = rt.n;
                        pack_n = 1;
                        pack_keys = len;
                        pack_rows = ql * G;
                        emit_run(e, 1, 0, rowbase, 0, 0);  // a pack of one is an ordinary unit
                    }
                } else {
                    pack_r = -1;
                    for (int ps = 0; ps < npass; ++ps) emit_run(e, nt, ps, rowbase, ql, 1);
                }
                rowbase += nt * ql;
              }
            }
            if (lane == 0) {
                hdr[0] = r;
                hdr[1] = 0;
                sched[0] = 0;
                for (int k = 0; k < NTICKET; ++k) sched[ticket_word(k)] = 0;
                if (par && rt.n <= rt.cap) {
                    sMeta[0] = r;
                    sMeta[1] = rt.n;
                    sMeta[4] = 1;
                } else if (!par) {
                    sMeta[4] = 0;
                    if (np) np_record_order(ul, r, Hkv, G, slots, chunk_c, hdr, rt);
                }
            }
            if (!par || rt.n <= rt.cap) break;
        }
    }
    __syncthreads();
    if (!sMeta[4]) return;
    const int NR = sMeta[1];
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int k = wave; k < NR; k += nwaves) {
        const int first = rt.r0[k], n = rt.nt[k], e = rT0[k], ps = rSp[k], prow0 = rProw[k], ql = rQl[k], aux = rAux[k];
        for (int j = lane; j < n; j += 64) {
            ul.src[first + j] = e;
            ul.aux[first + j] = aux > 0 ? j : aux;
            ul.pass[first + j] = ps;
            ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
            ul.prow[first + j] = prow0 + j * ql;
        }
    }
    if (!np) return;
    record_order_parallel(ul, rt, NR, rT0, rSp, sMeta, hdr, Hkv, G, slots, chunk_c);
}

__global__ __launch_bounds__(128) void node_records_kernel(const int64_t* node_kv, const int64_t* node_kv_offset,
                                                           const int64_t* node_kv_len, const int64_t* node_q,
                                                           const int64_t* node_q_offset, const int64_t* node_q_len, int G,
                                                           int rows, int64_t q_st, int64_t q_sh, int64_t kv_stride_slot,
                                                           UnitList ul, const int32_t* hdr, char* plan, int32_t* row_q,
                                                           const int32_t* cache_loc, int n_new, int64_t new_row_bytes,
                                                           int np) {
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    const int R = hdr[0];
    char* rec = plan + (int64_t)r * PLAN_BYTES;
    int64_t* ro = reinterpret_cast<int64_t*>(rec + PLAN_ROWOFF);
    uint32_t* mk = reinterpret_cast<uint32_t*>(rec + PLAN_MASK);
    int32_t* desc = reinterpret_cast<int32_t*>(rec + PLAN_DESC);
    if (r > R) {  // unused capacity (see flatten_records_kernel)
        if (k == 0) desc[4] = 0;
        return;
    }
    if (r == R) {  // sentinel
        ro[k] = 0;
        mk[k] = 0u;
        if (k == 0) {
            desc[0] = 0;
            desc[1] = 0;
            desc[2] = 1;
            desc[3] = -1;
            desc[4] = 0;
        }
        return;
    }
    const int u = np ? ul.perm[r] : r;  // unit packed into this record
    if (k == 0) {
        desc[4] = np ? ul.ch_n[r] : 0;
        desc[5] = np ? ul.ch_fb[r] : 0;
    }
    const int e0 = ul.src[u], aux = ul.aux[u], ps = ul.pass[u], prow = ul.prow[u];
    if (aux < 0) {
        // ---- packed unit: entries e0 .. e0 - aux - 1, each one tile and one pass -------------------------
        const int cnt = -aux;
        // slot k -> (entry j, position inside it); virtual row k -> (entry j, query, head of the group)
        int kj = -1, kpos = 0, kvb = 0, knv = 0;  // for the slot role
        int vj = -1, vloc = 0, vrowbase = 0;      // for the row role (k < MQ)
        int keys = 0, vrows = 0, rowsum = 0;
        int first_slot_e = e0;
        for (int j = 0; j < cnt; ++j) {
            const int e = e0 + j;
            const int len = (int)node_kv_len[e];
            const int nv = (int)node_q_len[e] * G;
            if (kj < 0 && k < keys + len) {
                kj = e;
                kpos = k - keys;
                kvb = vrows;
                knv = nv;
            }
            if (vj < 0 && k < vrows + nv) {
                vj = e;
                vloc = k - vrows;
                vrowbase = rowsum;
            }
            keys += len;
            vrows += nv;
            rowsum += (int)node_q_len[e];
        }
        const bool live = kj >= 0;
        const int64_t slot = live ? node_kv[node_kv_offset[kj] + kpos] : node_kv[node_kv_offset[first_slot_e]];
        ro[k] = plan_rowoff(slot, kv_stride_slot, cache_loc, n_new, new_row_bytes);
        mk[k] = live ? ((knv >= 32 ? 0xffffffffu : ((1u << knv) - 1u)) << kvb) : 0u;
        if (k < MQ) {
            int qs, orow = 0;
            if (vj >= 0) {
                const int qi = vloc / G, g = vloc % G;
                const int64_t qrow = node_q[node_q_offset[vj] + qi];
                qs = (int)(qrow * q_st + g * q_sh);
                orow = g * rows + prow + vrowbase + qi;
                if (g == 0) row_q[prow + vrowbase + qi] = (int32_t)qrow;
            } else {
                qs = (int)(node_q[node_q_offset[e0]] * q_st);  // alias a real row
            }
            reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = qs;
            reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = orow;
        }
        if (k == 0) {
            desc[0] = vrows;
            desc[1] = prow;
            desc[2] = 1;
            desc[3] = ul.flags[u] >> 1;
        }
        return;
    }
    const int e = e0, tt = aux;
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
        if (!np) {
            if (ps == 0 && k < ql) row_q[prow + k] = (int32_t)node_q[q0 + k];
        } else if (k < nv && (MQ * ps + k) % G == 0) {
            const int qi = (MQ * ps + k) / G;
            row_q[prow + qi] = ul.ch_n[r] > 0 ? (int32_t)node_q[q0 + qi] : -1;
        }
    }
    if (k == 0) {
        desc[0] = nv;
        desc[1] = prow;
        desc[2] = ul.flags[u] & 1;
        desc[3] = ul.flags[u] >> 1;
    }
}

}  // namespace deft
