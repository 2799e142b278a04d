// Window plan: the incremental half of the per-step head.
//
// The reference rebuilds all of TreeMetadata on every decode step (DeFT/deft/tree_decoding/tree_cache.py:619-881, timed at
// tree_generate.py:123-131), and so did the captured step of deft_amd.DecodeSession: upload, scan, blocks, units, records, row
// lists -- six launches of mostly one lone workgroup, ~58 us in front of every step's first layer.  The Flatten split cannot be
// patched in place: every live leaf grows by one token per step, so every block boundary behind the first leaf moves.  But the
// session owns its plan format, and inside a 128-slot tile the ORDER of the slots means nothing to attention (every slot carries
// its own row mask).  So a session plans for a WINDOW of steps:
//
//   * a REPLAN step runs the metadata and plan chain over the tree as it stands BEFORE the step's tokens -- the exact Flatten /
//     Node partition of that tree, no row flagged "new" -- and appends, per chunk c of max_q_len query rows (c < ceil(nq /
//     max_q_len)), `win_tiles` OVERFLOW tiles: blocks (Flatten) or one entry (Node) whose query list is rows [c * max_q_len, ...),
//     all masks zero.  The unit kernels keep an overflow run in ONE chunk (leader + followers, its own partial rows);
//   * every step -- the replan step included -- then runs window_patch_kernel, ONE workgroup: the journal replay and the advance
//     of the device tree (what tree_md_scan_kernel did), the page-table write, and a PATCH LIST the host made: {overflow region and
//     position, node, slot | new row}.  Every (chunk c, 32-row pass) pair is a REGION with overflow tiles of its own; a token
//     appended to leaf r lands in the one region that holds query r's rows, with the mask "query r only" -- computed from the
//     node's leaf set, so a slot merged into an inner node (speculative decoding's accepted tokens) gets that node's queries, in
//     every region that has one of them (with GQA a tile shared by all passes would be folded G times; a region's tile is folded
//     once, like a leaf tile of the exact plan); a RESET clears its node's positions, which the node's next tokens reuse.  The
//     leaders' tile count grows as tiles fill: a dormant tile costs nothing.
//   * when the overflow is full (or something happens the window cannot express) the host replays the REPLAN graph instead of the
//     PATCH graph.  deft_amd/session.py keeps the books; the device side below is deliberately dumb.
//
// Results differ from the eager path in the ORDER of fp32 additions only (another partition of the same keys): compared at the
// oracle's tolerance, not bit for bit.
//
// Included by deft_kernels.hip after tree_plan.h and plan_kernels.h.
#pragma once

namespace deft {

constexpr int WIN_ERR = 8;       // dims[TREE_ERR] bit: the overflow blocks do not fit the arrays' capacities

// Flatten: append ceil(nq / max_q_len) * win_tiles overflow blocks behind this step's blocks (dims[5], dims[6]).
__global__ __launch_bounds__(256) void window_blocks_kernel(TreeMdOut o, int32_t* dims, int nq, int max_q_len, int win_tiles,
                                                            int block_len, int nb_cap, int p_cap) {
    __shared__ int sNB, sP;
    if (threadIdx.x == 0) {
        sNB = dims[5];
        sP = dims[6];
    }
    __syncthreads();
    const int NB = sNB, P = sP;
    const int chunks = (nq + max_q_len - 1) / max_q_len;
    const int nblk = chunks * win_tiles;
    if (dims[TREE_ERR]) return;
    if (NB + nblk > nb_cap || P + win_tiles * nq > p_cap) {
        if (threadIdx.x == 0) atomicOr(dims + TREE_ERR, WIN_ERR);
        return;
    }
    for (int idx = 0; idx < nblk; ++idx) {
        const int c = idx / win_tiles, j = idx - c * win_tiles;
        const int cnt = min(max_q_len, nq - c * max_q_len);
        const int t = NB + idx;
        const int p0 = P + c * win_tiles * max_q_len + j * cnt;  // (every chunk in front of c is full)
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) o.block_q[p0 + i] = c * max_q_len + i;
        for (int k = threadIdx.x; k < block_len; k += blockDim.x) {
            o.block_kv[(int64_t)t * block_len + k] = 0;  // (a valid slot: masked rows are still requested)
            o.block_bitmasks[(int64_t)t * block_len + k] = 0;
        }
        if (threadIdx.x == 0) {
            o.block_q_cnts[t] = cnt;
            o.block_q_offset[t] = p0;
            o.block_lens[t] = block_len;
        }
    }
    if (threadIdx.x == 0) {
        dims[WIN_DIM_NB] = NB;
        dims[WIN_DIM_P] = P;
        dims[5] = NB + nblk;
        dims[6] = P + win_tiles * nq;
        dims[7] = (NB + nblk) * block_len;
    }
}

// Node: append one overflow entry of win_tiles * 128 slots per query chunk behind this step's entries (dims[1], dims[3], dims[4]).
__global__ __launch_bounds__(256) void window_entries_kernel(TreeMdOut o, int32_t* dims, int nq, int max_q_len, int win_tiles,
                                                             int ne_cap, int q_cap, int kv_cap) {
    __shared__ int sNE, sQ, sKV;
    if (threadIdx.x == 0) {
        sNE = dims[1];
        sQ = dims[3];
        sKV = dims[4];
    }
    __syncthreads();
    const int NE = sNE, Q = sQ, KV = sKV;
    const int chunks = (nq + max_q_len - 1) / max_q_len;
    const int len = win_tiles * TILE;
    if (dims[TREE_ERR]) return;
    if (NE + chunks > ne_cap || Q + nq > q_cap || KV + chunks * len > kv_cap) {
        if (threadIdx.x == 0) atomicOr(dims + TREE_ERR, WIN_ERR);
        return;
    }
    for (int c = 0; c < chunks; ++c) {
        const int cnt = min(max_q_len, nq - c * max_q_len);
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) o.node_q[Q + c * max_q_len + i] = c * max_q_len + i;
        for (int k = threadIdx.x; k < len; k += blockDim.x) o.node_kv[KV + (int64_t)c * len + k] = 0;
        if (threadIdx.x == 0) {
            o.node_q_len[NE + c] = cnt;
            o.node_kv_len[NE + c] = len;
            o.node_q_offset[NE + c] = Q + c * max_q_len;
            o.node_kv_offset[NE + c] = KV + (int64_t)c * len;
        }
    }
    if (threadIdx.x == 0) {
        dims[WIN_DIM_NB] = NE;
        dims[WIN_DIM_P] = Q;
        dims[1] = NE + chunks;
        dims[3] = Q + nq;
        dims[4] = KV + chunks * len;
    }
}

// OPTIONAL (DecodeSession(staging="kernel"); the default is one async copy in front of the step): the step's host-written words --
// slot numbers, page-table coordinates, journal, patch list -- FETCHED by the step's first kernel straight from pinned host memory.
// Built against the idle queue around a stand-alone hipMemcpyAsync (~20 us in front of its blit kernel, ~6 behind it in round 5's
// loop); most of that turned out to be the per-step EVENTS in the stream, and with those gone the two forms measure equal -- while the
// kernel form, in about one run in twenty, ran a whole decode loop 2.6 x slower (reads over PCIe from a kernel; never seen with the
// copy: profiles/r6_staging_kernel_vs_copy.txt).  The host writes a RING of slots ({used bytes, 12 bytes of padding, payload}); the
// kernel's slot is a device-side step counter modulo the ring, so that one captured launch serves every step while the host runs
// slots ahead (an event behind every fourth step keeps it from lapping the GPU).
struct StageFetch {
    const char* ring;   // pinned host memory, device-accessible; null = nothing to fetch
    char* dst;          // the session's device-side staging area (what every later kernel of the step reads)
    int32_t* counter;   // steps fetched so far
    int slot_bytes, ring_n;
};

// whole workgroup; ends with a barrier behind which every thread may read dst
__device__ inline void stage_fetch(const StageFetch& f) {
    if (!f.ring) return;
    const int s = *f.counter;
    const uintx4* src = reinterpret_cast<const uintx4*>(f.ring + (size_t)(s % f.ring_n) * (size_t)f.slot_bytes);
    uintx4* dst = reinterpret_cast<uintx4*>(f.dst);
    const int cap16 = f.slot_bytes / 16 - 1;  // payload chunks the slot can hold
    // header and first payload chunk of every thread in ONE round trip over PCIe (the chunk speculatively: it is inside the slot)
    const int tid = threadIdx.x, nth = blockDim.x;
    const uintx4 hdr = __builtin_nontemporal_load(src);
    uintx4 v = {0u, 0u, 0u, 0u};
    if (tid < cap16) v = __builtin_nontemporal_load(src + 1 + tid);
    const int used16 = min(cap16, (int)((hdr[0] + 15u) / 16u));
    if (tid < used16) dst[tid] = v;
    for (int i = tid + nth; i < used16; i += nth) dst[i] = __builtin_nontemporal_load(src + 1 + i);
    __threadfence();
    __syncthreads();
    if (tid == 0) *f.counter = s + 1;
}

__global__ __launch_bounds__(1024) void stage_fetch_kernel(StageFetch f) { stage_fetch(f); }

constexpr int WIN_REGIONS = 64;  // hosting records -- (query chunk, 32-row pass) pairs -- a window plan's tables have room for

struct WindowPatch {
    const int32_t* ops;        // journal {words, ...} to replay on the device tree first, or null (a replan step: the scan did)
    const int32_t* cache_loc;  // this step's slots: appended to the leaves of the device tree
    const int32_t* patch;      // {entries, active overflow tiles of region 0 .. 63, {region << 20 | position, row mask (0 = cleared),
                               //  slot | -1 - new row} ...} (window_host.cpp)
    const int32_t* tab;        // [chunk][WIN_PASSES][2]: leader record, first follower record (the record kernels write it)
    char* records;             // plan records
    int32_t* err;              // dims + TREE_ERR
    int max_q_len, G, win_tiles;
    int64_t kv_stride_slot;    // elements
    int64_t new_row_bytes;
};

// ONE workgroup of 1024 threads per decode step.  Region h of the overflow = the h-th hosting record, chunk by chunk, pass by pass
// (window_host.cpp hands out positions per region -- a leaf's token lands in the ONE region that holds its query's rows -- and
// computes every entry's row mask).
// A lone workgroup pays 1-2 us per dependent round trip, and this kernel stands between two steps' layers, so its chains run SIDE BY
// SIDE in different waves (s_waitcnt counts per wave): waves 0-7 bring the step's words over PCIe where the session asks for that
// (stage_fetch: counter, then header and payload in one round trip) while waves 8-15 walk the device tree -- leaf row -> node -> length, room, last slot: three
// dependent loads -- and read the overflow runs' record table; one barrier; then the stores.  (13 us -> 8 as one chain after the
// other, profiles/r6_step_boundary.txt.)
__global__ __launch_bounds__(1024) void window_patch_kernel(TreeDev t, WindowPatch w, PageWrite pw, StageFetch fetch) {
    __shared__ int sNew[TREE_OPS_NEW], sPos[TREE_OPS_NEW], sMeta[4], sOps[TREE_OPS_LDS];
    __shared__ int sHost[WIN_REGIONS * 2];  // hosting records: (chunk, pass) pairs
    __shared__ int sTab[WIN_REGIONS * 2];   // ... and their {leader record, first follower record}
    __shared__ int sNH;
    const int tid = threadIdx.x;
    const bool tree_side = tid >= 512;  // (wave-uniform)
    const int r0 = tid - 512;
    // ---- tree side, in front of the barrier: everything that does not depend on the step's words -------------------------
    int li = -1, llen = 0, lstart = 0, lroom = 0, llast = 0;
    if (tree_side) {
        if (r0 == 0) {
            const int chunks = (t.nq + w.max_q_len - 1) / w.max_q_len;
            int nh = 0;
            for (int c = 0; c < chunks; ++c) {
                const int cnt = min(w.max_q_len, t.nq - c * w.max_q_len);
                const int pc = (cnt * w.G + MQ - 1) / MQ;
                for (int ps = 0; ps < pc && ps < WIN_PASSES && nh < WIN_REGIONS; ++ps, ++nh) {
                    sHost[2 * nh] = c;
                    sHost[2 * nh + 1] = ps;
                    sTab[2 * nh] = w.tab[(c * WIN_PASSES + ps) * 2];
                    sTab[2 * nh + 1] = w.tab[(c * WIN_PASSES + ps) * 2 + 1];
                }
            }
            sNH = nh;
        }
        for (int r = r0; r < t.nq && r == r0; r += 512) {  // (the first 512 query rows ahead of time; more than that: below)
            li = t.leaf_node[r];
            llen = t.node_len[li];
            lstart = t.node_start[li];
            lroom = t.node_cap[li];
            llast = llen > 0 ? t.slots[lstart + llen - 1] : (int)0x80000000;
        }
    } else {
        // (waves 0-7 fetch; with nothing to fetch -- a caller that copied the words itself -- they only meet the barrier)
    }
    if (fetch.ring) {
        // the fetch's own barrier is the one both sides meet: every thread takes part (the tree side's threads copy too, behind
        // their loads -- harmless, the chunks are distributed over all 1024 threads)
        stage_fetch(fetch);
    } else {
        __syncthreads();
    }
    const int n_ent = w.patch[0];
    const int nops = w.ops ? w.ops[0] : 0;
    // ---- the device tree: journal, page table, this step's slots (tree_md_scan_kernel's opening) -----------------------
    if (nops > 0) {  // (uniform; speculative-decoding steps: the journal changes node lengths, the walk above is void)
        __syncthreads();
        tree_apply_ops(t, w.ops, w.err, sNew, sPos, sMeta, sOps);
    }
    if (pw.table)
        for (int r = tid; r < t.nq; r += 1024) pw.table[pw.rows[r] * pw.stride + pw.cols[r]] = w.cache_loc[r];
    if (tree_side) {
        for (int r = r0; r < t.nq; r += 512) {
            int i = li, len = llen, start = lstart, room = lroom, last = llast;
            if (nops > 0 || r != r0) {
                i = t.leaf_node[r];
                len = t.node_len[i];
                start = t.node_start[i];
                room = t.node_cap[i];
                last = len > 0 ? t.slots[start + len - 1] : (int)0x80000000;
            }
            if (len >= room) {
                atomicOr(w.err, 1);
                continue;
            }
            int32_t* sl = t.slots + start;
            const int32_t v = w.cache_loc[r];
            int p = len;
            if (last > v)  // (rare: a slot freed by a cut came back lower -- keep the node's list ascending)
                while (p > 0 && sl[p - 1] > v) {
                    sl[p] = sl[p - 1];
                    --p;
                }
            sl[p] = v;
            t.node_len[i] = len + 1;
        }
    }
    // ---- the plan: one thread per entry ------------------------------------------------------------------------------------
    const int NH = sNH;
    const int32_t* ent = w.patch + 1 + WIN_REGIONS;
    for (int e = tid; e < n_ent; e += 1024) {
        const int key = ent[3 * e], val = ent[3 * e + 2];
        const uint32_t mask = (uint32_t)ent[3 * e + 1];
        const int hi = key >> 20, pos = key & 0xfffff;
        const int j = pos >> 7, k = pos & (TILE - 1);
        if (hi < 0 || hi >= NH || j >= w.win_tiles) {
            atomicOr(w.err, WIN_ERR);
            continue;
        }
        const int rec = j == 0 ? sTab[2 * hi] : sTab[2 * hi + 1] + j - 1;
        char* rp = w.records + (int64_t)rec * PLAN_BYTES;
        const int64_t ro = val >= 0 ? (int64_t)val * w.kv_stride_slot * 2 : (((int64_t)1 << 63) | ((int64_t)(-1 - val) * w.new_row_bytes));
        reinterpret_cast<int64_t*>(rp + PLAN_ROWOFF)[k] = ro;
        reinterpret_cast<uint32_t*>(rp + PLAN_MASK)[k] = mask;
    }
    // the leaders' tile count: tiles that hold nothing yet are not visited (the leader's own tile always is)
    for (int hi = tid; hi < NH; hi += 1024)
        reinterpret_cast<int32_t*>(w.records + (int64_t)sTab[2 * hi] * PLAN_BYTES + PLAN_DESC)[4] = max(1, min(w.patch[1 + hi], w.win_tiles));
}

}  // namespace deft
