// Device-side TreeMetadata: the KV-guided grouping and the flattened split of
// TreeMetadata.from_tree_cache (DeFT/deft/tree_decoding/tree_cache.py:618-881) as HIP kernels over a compact tree that
// LIVES on the GPU, so that a decode step uploads nothing but its nq new slot numbers.
//
// Compact tree (laid out by the host-side tree, tree.cpp deft_tree_layout_*, uploaded when the STRUCTURE changes --
// branch / cut / merge -- not per step):
//   node_start / node_len / node_cap [n] int32   nodes in DFS pre-order; a node's pool slots, ascending, at
//                                                slots[node_start .. + node_len), with room for node_cap
//   refs [n][nqw] uint64                         bit r: query row r (the r-th live leaf by id) is below the node
//   leaf_node [nq] int32                         DFS index of query row r's leaf
// Per decode step: tree_advance_kernel appends this step's slots (cache_loc) to the leaves, then three kernels emit
// the reference's twelve int64 arrays, bit for bit what deft_md_build (host.cpp) returns for the same tree:
//   tree_md_scan_kernel    one workgroup: prefix sums over nodes and 128-slot blocks, sizes -> dims[]
//   tree_md_blocks_kernel  one workgroup per physical block: block_kv / block_bitmasks / block_q ... (flattened split,
//                          tree_cache.py:653-723, :763-799; a block whose nodes have more than max_q_len queries is
//                          emitted once per query chunk)
//   tree_md_nodes_kernel   one workgroup per node: node_q / node_kv ... (KV-guided grouping, :725-762)
//
// Included by deft_kernels.hip.
#pragma once

namespace deft {

struct TreeDev {
    int n, nq, nqw;
    const int32_t* node_start;
    int32_t* node_len;
    const int32_t* node_cap;
    const unsigned long long* refs;
    const int32_t* leaf_node;
    int32_t* slots;
};

struct TreeMdOut {  // the reference's arrays (tree_cache.py:813-857), int64 on the device, capacities checked by the host
    int64_t *node_q, *node_kv, *node_q_len, *node_kv_len, *node_q_offset, *node_kv_offset;
    int64_t *block_q, *block_q_cnts, *block_q_offset, *block_bitmasks, *block_kv, *block_lens;
};

// scratch of one build (int32 unless noted), carved by the host wrapper
struct TreeScratch {
    int32_t* pos;       // [n + 1]   flattened position of every node's first slot
    int32_t* e_off;     // [n + 1]   first Node entry of every node
    int32_t* q_off;     // [n + 1]   first node_q element
    int32_t* kv_off;    // [n + 1]   first node_kv element
    int32_t* b_first;   // [nbp]     first node that has a slot in the block
    int32_t* b_eoff;    // [nbp + 1] first emitted block of the physical block
    int32_t* b_poff;    // [nbp + 1] first block_q element
    unsigned long long* b_union;  // [nbp][nqw]
    int32_t* dims;      // [16] query_num, NE, total_kv, len(node_q), len(node_kv), NB, P, len(block_kv), physical blocks, error
};
constexpr int TREE_DIMS = 16;
constexpr int TREE_ERR = 9;

// one slot per live leaf, kept ascending inside the node (the pool hands out ascending slots, so the new one almost
// always goes to the end; after a cut freed slots come back lower and are inserted)
__global__ __launch_bounds__(256) void tree_advance_kernel(TreeDev t, const int32_t* cache_loc, int32_t* err) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.nq) return;
    const int i = t.leaf_node[r];
    const int len = t.node_len[i];
    if (len >= t.node_cap[i]) {
        atomicOr(err, 1);  // out of room: the host re-lays the tree out before this can happen
        return;
    }
    int32_t* s = t.slots + t.node_start[i];
    const int32_t v = cache_loc[r];
    int p = len;
    while (p > 0 && s[p - 1] > v) {
        s[p] = s[p - 1];
        --p;
    }
    s[p] = v;
    t.node_len[i] = len + 1;
}

// Replay of the host tree's journal (tree.cpp: changes a structural epoch absorbs) on the compact tree, by ONE workgroup of
// 1024 threads: ops[0] = words that follow; {1 = EXTEND, node, n, n slots ...} | {2 = RESET, node, 0}, oldest first.
//   RESET   the node's slots are dropped (its room stays)
//   EXTEND  n slots join the node's ascending slot list -- appended when they are all larger than the node's last slot (a pool
//           hands out ascending slots), otherwise merged in: the new slots are ranked among themselves in LDS, every existing
//           slot behind the first insertion point moves right by the number of new slots below it (chunks of 1024 from the END, so
//           nothing is overwritten before it is read), the new ones drop into the gaps.
// What the reference's speculative-decoding mock does every step -- the accepted leaves' slots squeezed into the root, every
// leaf's KV released (branch_func_example.py:420-437) -- is one EXTEND and nq RESETs, and the step stays inside its epoch.
constexpr int TREE_OPS_NEW = 1024;  // slots per EXTEND (longer ones are split by the host)
constexpr int TREE_OPS_LDS = 2048;  // journal words staged in LDS (longer journals are parsed where they lie)
__device__ inline void tree_apply_ops(const TreeDev& t, const int32_t* ops_g, int32_t* err, int* sNew, int* sPos, int* sMeta,
                                      int* sOps) {
    const int tid = threadIdx.x;
    const int words = ops_g[0];
    // the journal in LDS by one coalesced read: parsing it word by word in global memory was a dependent round trip per op
    const int32_t* ops = ops_g;
    if (words + 1 <= TREE_OPS_LDS) {
        for (int i = tid; i <= words; i += blockDim.x) sOps[i] = ops_g[i];
        __syncthreads();
        ops = sOps;
    }
    for (int at = 1; at + 2 < words + 1;) {  // (uniform: every thread reads the same words)
        const int op = ops[at], node = ops[at + 1], k = ops[at + 2];
        // (a RESET carries no slots: one with k != 0 would pass for "not a reset" below and the walk would never advance)
        if (node < 0 || node >= t.n || k < 0 || at + 3 + k > words + 1 || k > TREE_OPS_NEW || (op == 2 && k != 0)) {
            if (tid == 0) atomicOr(err, 4);
            break;
        }
        if (op == 2) {
            // a RUN of RESETs (a speculative-decoding step has one per leaf) in one go: thread j looks at the j-th op from here
            const int mine = at + 3 * tid;
            const bool is_reset = mine + 2 < words + 1 && ops[mine] == 2 && ops[mine + 2] == 0 && ops[mine + 1] >= 0 && ops[mine + 1] < t.n;
            if (tid == 0) sMeta[1] = 1024;
            __syncthreads();
            if (!is_reset) atomicMin(&sMeta[1], tid);
            __syncthreads();
            const int run = sMeta[1];  // >= 1: this op itself is a valid RESET
            if (tid < run) t.node_len[ops[mine + 1]] = 0;
            __threadfence_block();
            __syncthreads();
            at += 3 * run;
            continue;
        } else if (op == 1 && k > 0) {
            int32_t* s = t.slots + t.node_start[node];
            const int len = t.node_len[node];
            if (len + k > t.node_cap[node]) {
                if (tid == 0) atomicOr(err, 1);
            } else {
                // the new slots ascending: rank among themselves (slots are distinct; ties broken by index all the same)
                if (tid < k) {
                    const int v = ops[at + 3 + tid];
                    int r = 0;
                    for (int j = 0; j < k; ++j) {
                        const int u = ops[at + 3 + j];
                        r += (u < v || (u == v && j < tid)) ? 1 : 0;
                    }
                    sNew[r] = v;
                }
                __syncthreads();
                // where each new slot goes: (existing slots below it) + (new slots below it)
                if (tid < k) {
                    const int v = sNew[tid];
                    int lo = 0, hi = len;  // first existing slot > v
                    // (UPPER bound: a slot the node already holds -- merge_nodes(A, B, pruneB_flag=False) twice without the reset in
                    //  between hands A the same slot again, tree_cache.py:300-325 allows it -- goes BEHIND its twin; with the lower
                    //  bound the twin, which shifts by the new slots strictly below it, landed on the same position, one of the two was
                    //  lost and a stale word stayed in the list: a wild slot number in stage 1.  Round 5.)
                    // (a pool hands out ascending slots, so the new one usually goes behind the node's last: ONE load decides
                    //  that; the binary search is eleven dependent loads on a 1000-token node, 8 us of a decode step)
                    if (len == 0 || s[len - 1] <= v) lo = len;
                    else
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (s[mid] <= v) lo = mid + 1;
                            else hi = mid;
                        }
                    sPos[tid] = lo;
                    if (tid == 0) sMeta[0] = lo;  // first insertion point: nothing in front of it moves
                }
                __syncthreads();
                const int p = sMeta[0];
                for (int hi = len; hi > p;) {
                    const int lo = hi - 1024 > p ? hi - 1024 : p;
                    const int i = lo + tid;
                    int x = 0, shift = 0;
                    if (i < hi) {
                        x = s[i];
                        int a = 0, b = k;  // new slots below x
                        while (a < b) {
                            const int mid = (a + b) >> 1;
                            if (sNew[mid] < x) a = mid + 1;
                            else b = mid;
                        }
                        shift = a;
                    }
                    __syncthreads();
                    if (i < hi) s[i + shift] = x;
                    __syncthreads();
                    hi = lo;
                }
                if (tid < k) s[sPos[tid] + tid] = sNew[tid];
                if (tid == 0) t.node_len[node] = len + k;
            }
        }
        __threadfence_block();
        __syncthreads();
        at += 3 + k;
    }
}

__global__ __launch_bounds__(1024) void tree_ops_kernel(TreeDev t, const int32_t* ops, int32_t* err) {
    __shared__ int sNew[TREE_OPS_NEW], sPos[TREE_OPS_NEW], sMeta[4], sOps[TREE_OPS_LDS];
    tree_apply_ops(t, ops, err, sNew, sPos, sMeta, sOps);
}

// exclusive scan of f(i), i < m, into out[0 .. m] by the whole workgroup (1024 threads, chunks of 1024 with a carry)
// Returns the total (every thread): a caller that needs it does not read out[m] back -- with `out` in global memory that is a
// store and a dependent load, two round trips of a lone workgroup.
template <class F>
__device__ inline int block_exclusive_scan(int m, F f, int32_t* out, int* sWave, int* sCarry) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *sCarry = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int i = base + tid;
        const int v = i < m ? f(i) : 0;
        int inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(inc, d, 64);
            if (lane >= d) inc += u;
        }
        if (lane == 63) sWave[wave] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < wave; ++k) wbase += sWave[k];
        const int carry = *sCarry;
        if (i < m) out[i] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 1023) *sCarry = carry + wbase + inc;
        __syncthreads();
    }
    const int total = *sCarry;
    if (tid == 0) out[m] = total;
    __syncthreads();  // (everybody has read the carry before the next scan clears it)
    return total;
}

__device__ inline int refs_count(const unsigned long long* r, int nqw) {
    int c = 0;
    for (int w = 0; w < nqw; ++w) c += __popcll(r[w]);
    return c;
}

// The page-table rows of a step's new tokens (ReqToTokenPool.req_to_token[row[r], col[r]] = cache_loc[r], what TreeCache.alloc
// writes with one index_put, tree_cache.py:270-283): folded into the advance -- one launch fewer per captured step.
struct PageWrite {
    int32_t* table;        // null: nothing to write
    int64_t stride;        // elements per page-table row
    const int64_t* rows;   // [nq] request row of every query
    const int64_t* cols;   // [nq] position of its new token
};

// One workgroup.  `adv` (cache_loc != null): first append this step's slots to the leaves (tree_advance_kernel's work,
// folded in: one launch fewer per step).  The scans run over tables in LDS when the tree fits (TREE_LDS_NODES nodes,
// TREE_LDS_BLOCKS blocks: every tree but a pathological one) -- a scan whose input and output live in global memory pays
// an L2 round trip per phase, seven phases -- and the results are written out once at the end.
constexpr int TREE_LDS_NODES = 4096;
constexpr int TREE_LDS_BLOCKS = 8192;

__global__ __launch_bounds__(1024) void tree_md_scan_kernel(TreeDev t, TreeScratch s, int max_q_len, int block_len,
                                                            int max_block_len, int nbp_cap, const int32_t* cache_loc,
                                                            const int32_t* ops, PageWrite pw) {
    __shared__ int sWave[16];
    __shared__ int sCarry;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = t.n, nqw = t.nqw;
    const int tid = threadIdx.x;
    const bool in_lds = n <= TREE_LDS_NODES;
    // tables: [n+1] each -- LDS when the tree fits, else the global scratch directly
    int32_t* pos = in_lds ? reinterpret_cast<int32_t*>(smem) : s.pos;
    int32_t* e_off = in_lds ? pos + (n + 1) : s.e_off;
    int32_t* q_off = in_lds ? e_off + (n + 1) : s.q_off;
    int32_t* kv_off = in_lds ? q_off + (n + 1) : s.kv_off;
    int32_t* nlen = in_lds ? kv_off + (n + 1) : nullptr;   // node_len copy
    int32_t* nqc = in_lds ? nlen + (n + 1) : nullptr;      // leaf count per node
    // Round 5: every load that depends on nothing goes out FIRST and together -- the journal's word count, the node table (lengths,
    // leaf sets), this step's leaves and slots -- instead of one after the other behind the branch on the word count and behind the
    // advance's fence: a lone workgroup pays 1-2 us per dependent round trip, and this kernel opened with four of them.
    const int nops = ops ? ops[0] : 0;
    auto load_tables = [&]() {
        for (int i = tid; i < n; i += 1024) {
            nlen[i] = t.node_len[i];
            nqc[i] = refs_count(t.refs + (size_t)i * nqw, nqw);
        }
    };
    if (in_lds) load_tables();
    if (nops > 0) {  // the journal of this step's absorbed changes first (uniform branch), then the step's new slots
        __shared__ int sNew[TREE_OPS_NEW], sPos[TREE_OPS_NEW], sMeta[4], sOps[TREE_OPS_LDS];
        __syncthreads();
        tree_apply_ops(t, ops, s.dims + TREE_ERR, sNew, sPos, sMeta, sOps);
        if (in_lds) load_tables();  // (the replay changed node lengths: read them again -- speculative-decoding steps only)
    }
    if (cache_loc && pw.table)
        for (int r = tid; r < t.nq; r += 1024) pw.table[pw.rows[r] * pw.stride + pw.cols[r]] = cache_loc[r];
    if (in_lds) __syncthreads();  // the tables are in the LDS
    if (cache_loc) {  // advance: one slot per live leaf, kept ascending inside the node
        for (int r = tid; r < t.nq; r += 1024) {
            const int i = t.leaf_node[r];
            const int len = in_lds ? nlen[i] : t.node_len[i];
            if (len >= t.node_cap[i]) {
                atomicOr(s.dims + TREE_ERR, 1);
                continue;
            }
            int32_t* sl = t.slots + t.node_start[i];
            const int32_t v = cache_loc[r];
            int p = len;
            while (p > 0 && sl[p - 1] > v) {
                sl[p] = sl[p - 1];
                --p;
            }
            sl[p] = v;
            t.node_len[i] = len + 1;
            if (in_lds) nlen[i] = len + 1;  // (a leaf is one query row: nobody else touches its entry)
        }
        if (!in_lds) __threadfence_block();
        __syncthreads();
    }
    auto len_of = [&](int i) { return in_lds ? nlen[i] : t.node_len[i]; };
    auto nq_of = [&](int i) { return in_lds ? nqc[i] : refs_count(t.refs + (size_t)i * nqw, nqw); };
    auto qch = [&](int i) { return (nq_of(i) + max_q_len - 1) / max_q_len; };
    auto kch = [&](int i) {
        const int len = len_of(i);
        const int step = max_block_len == -1 ? len : max_block_len;
        return step > 0 ? (len + step - 1) / step : 0;
    };
    const int total = block_exclusive_scan(n, [&](int i) { return len_of(i); }, pos, sWave, &sCarry);
    const int tot_e = block_exclusive_scan(n, [&](int i) { return qch(i) * kch(i); }, e_off, sWave, &sCarry);
    const int tot_q = block_exclusive_scan(n, [&](int i) { return nq_of(i) * kch(i); }, q_off, sWave, &sCarry);
    const int tot_kv = block_exclusive_scan(n, [&](int i) { return len_of(i) * qch(i); }, kv_off, sWave, &sCarry);
    const int nbp = (total + block_len - 1) / block_len;
    if (nbp > nbp_cap) {
        if (tid == 0) {
            s.dims[8] = nbp;
            atomicOr(s.dims + TREE_ERR, 2);
        }
        return;
    }
    const bool blk_lds = in_lds && nbp_cap <= TREE_LDS_BLOCKS;  // (as the host sized the LDS: by capacity, not by this step's count)
    int32_t* b_cnt = blk_lds ? nqc + (n + 1) : nullptr;  // [nbp] union size per block
    // physical blocks: first node with a slot in the block (largest i with pos[i] <= lo among nodes that have slots),
    // union of the leaf sets of its nodes
    // (sixteen lanes per block: the nodes of a block are OR-ed together by the lanes in parallel -- one thread per block walked
    //  them one global load after the other, 64 of them for the block of one-token leaves of a Medusa-64 step: 15 us of a step)
    {
        const int sub = tid & 15, grp = tid >> 4;  // 64 groups of 16 lanes
        for (int b = grp; b < nbp; b += 64) {
            const int lo = b * block_len, hi = min(total, lo + block_len);
            int a = 0, z = n - 1;  // largest i with pos[i] <= lo
            while (a < z) {
                const int mid = (a + z + 1) >> 1;
                if (pos[mid] <= lo) a = mid;
                else z = mid - 1;
            }
            while (a < n && pos[a + 1] <= lo) ++a;  // (skip empty nodes that share the position)
            int e = a + 1, ze = n;  // first node behind a whose first position is >= hi (n if none)
            while (e < ze) {
                const int mid = (e + ze) >> 1;
                if (pos[mid] >= hi) ze = mid;
                else e = mid + 1;
            }
            unsigned long long* u = s.b_union + (size_t)b * nqw;
            int cnt = 0;
            for (int w = 0; w < nqw; ++w) {
                unsigned long long acc = 0ull;
                for (int j = a + sub; j < e; j += 16)
                    if (len_of(j) > 0) acc |= t.refs[(size_t)j * nqw + w];
                for (int m = 8; m > 0; m >>= 1) acc |= __shfl_xor(acc, m, 16);
                if (sub == 0) u[w] = acc;
                cnt += __popcll(acc);
            }
            if (sub == 0) {
                s.b_first[b] = a;
                if (blk_lds) b_cnt[b] = cnt;
            }
        }
    }
    __syncthreads();
    auto bcount = [&](int b) { return blk_lds ? b_cnt[b] : refs_count(s.b_union + (size_t)b * nqw, nqw); };
    const int tot_be = block_exclusive_scan(nbp, [&](int b) { return (bcount(b) + max_q_len - 1) / max_q_len; }, s.b_eoff, sWave, &sCarry);
    const int tot_bp = block_exclusive_scan(nbp, [&](int b) { return bcount(b); }, s.b_poff, sWave, &sCarry);
    if (in_lds)  // the node tables leave the LDS in one pass
        for (int i = tid; i <= n; i += 1024) {
            s.pos[i] = pos[i];
            s.e_off[i] = e_off[i];
            s.q_off[i] = q_off[i];
            s.kv_off[i] = kv_off[i];
        }
    if (tid == 0) {
        s.dims[0] = t.nq;
        s.dims[1] = tot_e;
        s.dims[2] = total;
        s.dims[3] = tot_q;
        s.dims[4] = tot_kv;
        s.dims[5] = tot_be;
        s.dims[6] = tot_bp;
        s.dims[7] = tot_be * block_len;
        s.dims[8] = nbp;
    }
}

// k-th (0-based) set bit of a bit set, or -1
__device__ inline int nth_set_bit(const unsigned long long* u, int nqw, int k) {
    for (int w = 0; w < nqw; ++w) {
        const int c = __popcll(u[w]);
        if (k < c) {
            unsigned long long x = u[w];
            for (int j = 0; j < k; ++j) x &= x - 1;
            return 64 * w + __ffsll((long long)x) - 1;
        }
        k -= c;
    }
    return -1;
}
// number of set bits of u below position q
__device__ inline int rank_below(const unsigned long long* u, int q) {
    int c = 0;
    for (int w = 0; w < (q >> 6); ++w) c += __popcll(u[w]);
    return c + __popcll(u[q >> 6] & ((1ull << (q & 63)) - 1ull));
}

// The masks of a block's positions depend on the position's NODE only (bit = rank of a leaf of the node among the block's
// query list, per chunk of max_q_len): they are computed once per (node of the block, chunk) into an LDS table -- a thread per
// node -- and every position looks its node's up.  (Computing them per position repeated the walk over the node's leaf set 128
// times: 18 us per step on a Medusa-64 tree, whose root's 64 leaves are ranked for every one of its positions.)
constexpr int BLK_TAB_NODES = 128, BLK_TAB_CHUNKS = 8;

__global__ __launch_bounds__(128) void tree_md_blocks_kernel(TreeDev t, TreeScratch s, TreeMdOut o, int max_q_len, int block_len) {
    __shared__ long long sMask[BLK_TAB_NODES * BLK_TAB_CHUNKS];
    __shared__ int sPosN[BLK_TAB_NODES + 1];  // first flattened position of the block's nodes (and of the node behind them)
    const int b = blockIdx.x;
    if (s.dims[TREE_ERR] || b >= s.dims[8]) return;
    const int nqw = t.nqw, n = t.n;
    const int total = s.dims[2];
    const int lo = b * block_len, cur_len = min(block_len, total - lo);
    // (the block's query list as a bit set, staged in LDS when it is short: every rank below reads it)
    __shared__ unsigned long long sUni[16];
    const unsigned long long* uni = s.b_union + (size_t)b * nqw;
    if (nqw <= 16) {
        if ((int)threadIdx.x < nqw) sUni[threadIdx.x] = uni[threadIdx.x];
        __syncthreads();
        uni = sUni;
    }
    const int nqs = refs_count(uni, nqw);
    const int chunks = (nqs + max_q_len - 1) / max_q_len;
    const int e0 = s.b_eoff[b], p0 = s.b_poff[b];
    // mask of node j for chunk c: rows of the chunk = union ranks [c * max_q_len, (c + 1) * max_q_len)
    auto node_mask = [&](int j, int c) {
        long long mask = 0;
        const unsigned long long* rj = t.refs + (size_t)j * nqw;
        // a node whose leaves are the whole list (a shared-prefix block): every row of the chunk, no ranking
        bool all = true;
        for (int w = 0; w < nqw; ++w) all &= (rj[w] & uni[w]) == uni[w];
        if (all) {
            const int cnt = min(max_q_len, nqs - c * max_q_len);
            return cnt >= 64 ? (long long)-1 : (long long)((1ull << cnt) - 1ull);
        }
        for (int w = 0; w < nqw; ++w) {
            unsigned long long x = rj[w];
            while (x) {
                const int q = 64 * w + __ffsll((long long)x) - 1;
                x &= x - 1;
                const int rk = rank_below(uni, q) - c * max_q_len;
                if (rk >= 0 && rk < max_q_len) mask |= (long long)1 << rk;
            }
        }
        return mask;
    };
    // the nodes that have a position in this block: j0 .. j0 + nn - 1 (empty nodes in between included: never looked up)
    const int j0 = s.b_first[b];
    // (the last of them from the NEXT block's first node -- it either straddles the boundary or starts right behind it --
    //  instead of a walk over pos[], one dependent load per node)
    int j1 = n - 1;
    if (b + 1 < s.dims[8]) {
        const int jn = s.b_first[b + 1];
        j1 = s.pos[jn] < lo + cur_len ? jn : jn - 1;
    }
    if (j1 < j0) j1 = j0;
    const int nn = j1 - j0 + 1;
    const bool tab = nn <= BLK_TAB_NODES && chunks <= BLK_TAB_CHUNKS;
    if (tab) {
        for (int x = threadIdx.x; x < nn * chunks; x += blockDim.x) {
            const int jj = x / chunks, c = x - jj * chunks;
            sMask[jj * BLK_TAB_CHUNKS + c] = node_mask(j0 + jj, c);
        }
        for (int x = threadIdx.x; x <= nn; x += blockDim.x) sPosN[x] = s.pos[j0 + x];  // (pos has n + 1 entries)
        __syncthreads();
    }
    for (int k = threadIdx.x; k < block_len; k += blockDim.x) {
        int64_t slot = -1;
        int j = -1;
        if (k < cur_len) {
            if (tab) {
                // the LAST node of the block whose first position is <= lo + k (empty nodes share a position with their
                // successor and are skipped that way): a search in LDS -- walking pos[] node by node in global memory was one
                // dependent load per node, 64 of them for a block of one-token leaves
                int a = 0, z = nn - 1;
                while (a < z) {
                    const int mid = (a + z + 1) >> 1;
                    if (sPosN[mid] <= lo + k) a = mid;
                    else z = mid - 1;
                }
                j = j0 + a;
                slot = t.slots[t.node_start[j] + (lo + k - sPosN[a])];
            } else {
                j = j0;
                while (j + 1 < n && s.pos[j + 1] <= lo + k) ++j;  // (a block touches few nodes; empty nodes are skipped)
                slot = t.slots[t.node_start[j] + (lo + k - s.pos[j])];
            }
        }
        for (int c = 0; c < chunks; ++c) {
            int64_t mask = 0;
            if (j >= 0) mask = tab ? sMask[(j - j0) * BLK_TAB_CHUNKS + c] : node_mask(j, c);
            o.block_kv[(int64_t)(e0 + c) * block_len + k] = slot;
            o.block_bitmasks[(int64_t)(e0 + c) * block_len + k] = mask;
        }
    }
    for (int c = 0; c < chunks; ++c) {
        const int cnt = min(max_q_len, nqs - c * max_q_len);
        for (int k = threadIdx.x; k < cnt; k += blockDim.x)
            o.block_q[p0 + c * max_q_len + k] = nth_set_bit(uni, nqw, c * max_q_len + k);
        if (threadIdx.x == 0) {
            o.block_q_cnts[e0 + c] = cnt;
            o.block_q_offset[e0 + c] = p0 + c * max_q_len;
            o.block_lens[e0 + c] = cur_len;
        }
    }
}

// grid (nodes, parts): the parts of a row share a node's slot copy
__global__ __launch_bounds__(256) void tree_md_nodes_kernel(TreeDev t, TreeScratch s, TreeMdOut o, int max_q_len, int max_block_len) {
    const int i = blockIdx.x;
    if (s.dims[TREE_ERR] || i >= t.n) return;
    const int nqw = t.nqw;
    const unsigned long long* ri = t.refs + (size_t)i * nqw;
    const int nq_i = refs_count(ri, nqw);
    const int len = t.node_len[i];
    if (len <= 0 || nq_i <= 0) return;
    const int step = max_block_len == -1 ? len : max_block_len;
    const int kchunks = (len + step - 1) / step, qchunks = (nq_i + max_q_len - 1) / max_q_len;
    const int e0 = s.e_off[i], q0 = s.q_off[i], kv0 = s.kv_off[i];
    const int32_t* sl = t.slots + t.node_start[i];
    const int tid = blockIdx.y * blockDim.x + threadIdx.x, nth = gridDim.y * blockDim.x;
    // entries: q chunks outer, KV chunks inner (tree_cache.py:744-758)
    for (int e = tid; e < qchunks * kchunks; e += nth) {
        const int qc = e / kchunks, kc = e - qc * kchunks;
        const int qcnt = min(max_q_len, nq_i - qc * max_q_len), klen = min(step, len - kc * step);
        o.node_q_len[e0 + e] = qcnt;
        o.node_kv_len[e0 + e] = klen;
        o.node_q_offset[e0 + e] = q0 + qc * kchunks * max_q_len + kc * qcnt;
        o.node_kv_offset[e0 + e] = kv0 + qc * len + kc * step;
    }
    // node_q: per entry its q chunk
    const int nqel = nq_i * kchunks;
    for (int x = tid; x < nqel; x += nth) {
        // element x of the node's node_q stretch: full q chunks take kchunks * max_q_len elements each
        const int per_full = kchunks * max_q_len;
        int qc = x / per_full;
        int rem = x - qc * per_full;
        if (qc >= qchunks) {  // (only when the last chunk is full too)
            qc = qchunks - 1;
            rem = x - qc * per_full;
        }
        const int qcnt = min(max_q_len, nq_i - qc * max_q_len);
        const int within = rem % qcnt;
        o.node_q[q0 + x] = nth_set_bit(ri, nqw, qc * max_q_len + within);
    }
    // node_kv: the node's slots once per q chunk
    const int64_t nkv = (int64_t)len * qchunks;
    for (int64_t x = tid; x < nkv; x += nth) o.node_kv[kv0 + x] = sl[x % len];
}

}  // namespace deft
