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

constexpr int WIN_REGIONS = 64;  // hosting records -- (query chunk, 32-row pass) pairs -- a window plan's tables have room for

struct WindowPatch {
    const int32_t* ops;        // journal {words, ...} to replay on the device tree first, or null (a replan step: the scan did)
    const int32_t* cache_loc;  // this step's slots: appended to the leaves of the device tree
    const int32_t* patch;      // {entries, active overflow tiles of region 0 .. 63, {region << 20 | position, node | -1 = clear,
                               //  slot | -1 - new row} ...} (window_host.cpp)
    const int32_t* tab;        // [chunk][WIN_PASSES][2]: leader record, first follower record (the record kernels write it)
    char* records;             // plan records
    int32_t* err;              // dims + TREE_ERR
    int max_q_len, G, win_tiles;
    int64_t kv_stride_slot;    // elements
    int64_t new_row_bytes;
};

// ONE workgroup of 1024 threads per decode step.  Region h of the overflow = the h-th hosting record, chunk by chunk, pass by pass
// (window_host.cpp hands out positions per region: a leaf's token lands in the ONE region that holds its query's rows).
__global__ __launch_bounds__(1024) void window_patch_kernel(TreeDev t, WindowPatch w, PageWrite pw) {
    __shared__ int sNew[TREE_OPS_NEW], sPos[TREE_OPS_NEW], sMeta[4], sOps[TREE_OPS_LDS];
    __shared__ int sHost[WIN_REGIONS * 2];  // hosting records: (chunk, pass) pairs
    __shared__ int sNH;
    const int tid = threadIdx.x;
    // the patch list's loads first: they depend on nothing
    const int n_ent = w.patch[0];
    const int chunks = (t.nq + w.max_q_len - 1) / w.max_q_len;
    if (tid == 0) {
        int nh = 0;
        for (int c = 0; c < chunks; ++c) {
            const int cnt = min(w.max_q_len, t.nq - c * w.max_q_len);
            const int pc = (cnt * w.G + MQ - 1) / MQ;
            for (int ps = 0; ps < pc && ps < WIN_PASSES && nh < WIN_REGIONS; ++ps, ++nh) {
                sHost[2 * nh] = c;
                sHost[2 * nh + 1] = ps;
            }
        }
        sNH = nh;
    }
    // ---- the device tree: journal, page table, this step's slots (tree_md_scan_kernel's opening) -----------------------
    if (w.ops && w.ops[0] > 0) {
        __syncthreads();
        tree_apply_ops(t, w.ops, w.err, sNew, sPos, sMeta, sOps);
    }
    if (pw.table)
        for (int r = tid; r < t.nq; r += 1024) pw.table[pw.rows[r] * pw.stride + pw.cols[r]] = w.cache_loc[r];
    for (int r = tid; r < t.nq; r += 1024) {
        const int i = t.leaf_node[r];
        const int len = t.node_len[i];
        if (len >= t.node_cap[i]) {
            atomicOr(w.err, 1);
            continue;
        }
        int32_t* sl = t.slots + t.node_start[i];
        const int32_t v = w.cache_loc[r];
        int p = len;
        while (p > 0 && sl[p - 1] > v) {
            sl[p] = sl[p - 1];
            --p;
        }
        sl[p] = v;
        t.node_len[i] = len + 1;
    }
    __syncthreads();
    // ---- the plan: one thread per entry ------------------------------------------------------------------------------------
    const int NH = sNH;
    const int32_t* ent = w.patch + 1 + WIN_REGIONS;
    for (int e = tid; e < n_ent; e += 1024) {
        const int key = ent[3 * e], node = ent[3 * e + 1], val = ent[3 * e + 2];
        const int hi = key >> 20, pos = key & 0xfffff;
        const int j = pos >> 7, k = pos & (TILE - 1);
        if (hi < 0 || hi >= NH || j >= w.win_tiles) {
            atomicOr(w.err, WIN_ERR);
            continue;
        }
        const int c = sHost[2 * hi], ps = sHost[2 * hi + 1];
        const int32_t* tb = w.tab + (c * WIN_PASSES + ps) * 2;
        const int rec = j == 0 ? tb[0] : tb[1] + j - 1;
        char* rp = w.records + (int64_t)rec * PLAN_BYTES;
        uint32_t mask = 0u;
        int64_t ro = 0;
        if (node >= 0) {
            const unsigned long long* rf = t.refs + (size_t)node * t.nqw;
            const int cnt = min(w.max_q_len, t.nq - c * w.max_q_len);
            for (int v = 0; v < MQ; ++v) {
                const int qi = (MQ * ps + v) / w.G;
                const int q = c * w.max_q_len + qi;
                if (qi < cnt && ((rf[q >> 6] >> (q & 63)) & 1ull)) mask |= 1u << v;
            }
            ro = val >= 0 ? (int64_t)val * w.kv_stride_slot * 2 : (((int64_t)1 << 63) | ((int64_t)(-1 - val) * w.new_row_bytes));
        }
        reinterpret_cast<int64_t*>(rp + PLAN_ROWOFF)[k] = ro;
        reinterpret_cast<uint32_t*>(rp + PLAN_MASK)[k] = mask;
    }
    // the leaders' tile count: tiles that hold nothing yet are not visited (the leader's own tile always is)
    for (int hi = tid; hi < NH; hi += 1024) {
        const int32_t* tb = w.tab + (sHost[2 * hi] * WIN_PASSES + sHost[2 * hi + 1]) * 2;
        reinterpret_cast<int32_t*>(w.records + (int64_t)tb[0] * PLAN_BYTES + PLAN_DESC)[4] = max(1, min(w.patch[1 + hi], w.win_tiles));
    }
}

}  // namespace deft
