// Host-side books of a window plan (window.h; deft_amd.DecodeSession): which overflow position holds which node's slot, and the
// patch list of every decode step.  Pure host code, no HIP.
//
// The overflow has one REGION per hosting record of the plan -- (chunk c of max_q_len query rows, 32-row pass ps of its cnt x G
// virtual rows) -- of `tiles` x 128 positions each, handed out in order.  A slot of node i goes to every region that holds a row of
// a query below i: a leaf's token to exactly ONE (so an overflow tile is folded by one pass, like a leaf tile of the exact plan --
// with GQA a tile shared by all regions would be folded G times), a slot merged into the root to all of them.  A node that is RESET
// keeps the positions it was given and its next slots reuse them (a speculative-decoding leaf is reset and refilled on every step, branch_func_example.py:420-437), so
// such a window lasts until the slots MERGED into inner nodes fill it.  A step that the window cannot express -- the overflow is
// full; a node is reset whose slots the static part of the plan still holds -- returns -1 and the caller runs a REPLAN step
// (`replan` = 1), which starts the books afresh: the journal of that step is replayed by the metadata kernels BEFORE the plan is
// made, so its RESETs mark nodes "clean" (every slot of theirs, from now on, is in the overflow) and nothing of it is patched.
#include <algorithm>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace deft {

constexpr int WIN_REGIONS = 64;  // (window.h: hosting records a window plan's tables have room for)

struct WindowBooks {
    int nq = 0, n_nodes = 0, cap = 0, max_entries = 0, regions = 0;
    std::vector<int32_t> leaf_node;            // query row -> DFS index of its leaf
    std::vector<uint64_t> node_regions;        // node -> bit h: a query below the node has a row in region h
    std::vector<std::vector<uint32_t>> node_masks;  // node -> per set bit of node_regions (ascending h): bit v = virtual row v of region h belongs to a query below the node
    std::vector<std::vector<int32_t>> own;     // node -> overflow positions (region << 20 | position) it has been given, in order
    std::vector<int32_t> live;                 // node -> how many of them hold a slot
    std::vector<uint8_t> clean;                // node: none of its slots is in the static part of the plan
    int fill[WIN_REGIONS] = {};
    bool valid = false;
    struct New {
        int32_t pos, node, slot;
    };
    std::vector<New> prev_new;                 // the rows the LAST step read from k_new / v_new
    // this step's patch entries: last write to a position wins
    std::vector<int32_t> ent_node /* the entry's row mask */, ent_val, stamp, touched;
    int32_t step_no = 0;
    std::mutex mu;

    // one more slot of `node`: a position in every region of the node (reused ones first); false = some region is full
    bool place(int node, int val) {
        auto& o = own[(size_t)node];
        int n = live[(size_t)node];
        int ord = 0;
        for (uint64_t m = node_regions[(size_t)node]; m; m &= m - 1, ++ord) {
            const int h = __builtin_ctzll(m);
            int key;
            if (n < (int)o.size()) {
                key = o[(size_t)n];  // (positions are given region by region, in this order: the n-th is region h's again)
            } else {
                if (fill[h] >= cap) return false;
                key = (h << 20) | fill[h]++;
                o.push_back(key);
            }
            ++n;
            put(key, (int32_t)node_masks[(size_t)node][(size_t)ord], val);
        }
        live[(size_t)node] = n;
        return true;
    }
    // the row mask of `node` in the region of `key`
    int32_t mask_of(int key, int node) const {
        const int h = key >> 20;
        const uint64_t below = node_regions[(size_t)node] & ((1ull << h) - 1ull);
        return (int32_t)node_masks[(size_t)node][(size_t)__builtin_popcountll(below)];
    }
    void put(int key, int32_t mask, int val) {
        const size_t at = (size_t)(key >> 20) * (size_t)cap + (size_t)(key & 0xfffff);
        if (stamp[at] != step_no) {
            stamp[at] = step_no;
            touched.push_back(key);
        }
        ent_node[at] = mask;
        ent_val[at] = val;
    }
};

static std::mutex g_win_mu;
static std::unordered_map<int64_t, std::shared_ptr<WindowBooks>> g_windows;
static int64_t g_win_next = 1;

static std::shared_ptr<WindowBooks> find_window(int64_t h) {
    std::lock_guard<std::mutex> lk(g_win_mu);
    auto it = g_windows.find(h);
    return it == g_windows.end() ? nullptr : it->second;
}

}  // namespace deft

using namespace deft;

extern "C" {

int64_t deft_window_create(int n_nodes, int nq, int nqw, int tiles, const int32_t* leaf_node, const uint64_t* refs, int max_q_len,
                           int group, int max_entries) {
    if (n_nodes <= 0 || nq <= 0 || nqw < (nq + 63) / 64 || tiles < 1 || tiles * DEFT_BLOCK_LEN >= (1 << 20) || !leaf_node || !refs ||
        max_q_len < 1 || group < 1 || max_entries < nq) {
        set_error("deft_window_create: bad arguments (nodes=%d nq=%d tiles=%d entries=%d)", n_nodes, nq, tiles, max_entries);
        return DEFT_EINVAL;
    }
    auto w = std::make_shared<WindowBooks>();
    w->nq = nq;
    w->n_nodes = n_nodes;
    w->cap = tiles * DEFT_BLOCK_LEN;
    w->max_entries = max_entries;
    w->leaf_node.assign(leaf_node, leaf_node + nq);
    // regions in window_patch_kernel's order: chunk by chunk, pass by pass; query row q = c * max_q_len + qi has its `group` virtual
    // rows qi * group ... in passes (qi * group) / 32 ... (qi * group + group - 1) / 32 of chunk c
    std::vector<uint64_t> row_regions((size_t)nq, 0);
    int h0 = 0;
    for (int c = 0; c * max_q_len < nq; ++c) {
        const int cnt = std::min(max_q_len, nq - c * max_q_len);
        const int passes = (cnt * group + DEFT_MAX_Q_LEN - 1) / DEFT_MAX_Q_LEN;
        if (h0 + passes > WIN_REGIONS) {
            set_error("deft_window_create: more than %d hosting records", WIN_REGIONS);
            return DEFT_EUNSUPPORTED;
        }
        for (int qi = 0; qi < cnt; ++qi)
            for (int ps = qi * group / DEFT_MAX_Q_LEN; ps <= (qi * group + group - 1) / DEFT_MAX_Q_LEN; ++ps)
                row_regions[(size_t)(c * max_q_len + qi)] |= 1ull << (h0 + ps);
        h0 += passes;
    }
    w->regions = h0;
    w->node_regions.assign((size_t)n_nodes, 0);
    w->node_masks.assign((size_t)n_nodes, {});
    auto below = [&](int i, int q) { return ((refs[(size_t)i * (size_t)nqw + (size_t)(q >> 6)] >> (q & 63)) & 1ull) != 0; };
    for (int i = 0; i < n_nodes; ++i) {
        for (int q = 0; q < nq; ++q)
            if (below(i, q)) w->node_regions[(size_t)i] |= row_regions[(size_t)q];
        // virtual row v of region (c, ps) is head (32 ps + v) % group of query c * max_q_len + (32 ps + v) / group
        int hh = 0;
        for (int c = 0; c * max_q_len < nq; ++c) {
            const int cnt = std::min(max_q_len, nq - c * max_q_len);
            const int passes = (cnt * group + DEFT_MAX_Q_LEN - 1) / DEFT_MAX_Q_LEN;
            for (int ps = 0; ps < passes; ++ps, ++hh) {
                if (!((w->node_regions[(size_t)i] >> hh) & 1ull)) continue;
                uint32_t mask = 0;
                for (int v = 0; v < DEFT_MAX_Q_LEN; ++v) {
                    const int qi = (DEFT_MAX_Q_LEN * ps + v) / group;
                    if (qi < cnt && below(i, c * max_q_len + qi)) mask |= 1u << v;
                }
                w->node_masks[(size_t)i].push_back(mask);
            }
        }
    }
    for (int r = 0; r < nq; ++r)
        if (leaf_node[r] < 0 || leaf_node[r] >= n_nodes) {
            set_error("deft_window_create: leaf row %d names node %d of %d", r, leaf_node[r], n_nodes);
            return DEFT_EINVAL;
        }
    w->own.resize((size_t)n_nodes);
    w->live.assign((size_t)n_nodes, 0);
    w->clean.assign((size_t)n_nodes, 0);
    w->ent_node.assign((size_t)w->cap * (size_t)w->regions, 0);
    w->ent_val.assign((size_t)w->cap * (size_t)w->regions, 0);
    w->stamp.assign((size_t)w->cap * (size_t)w->regions, -1);
    std::lock_guard<std::mutex> lk(g_win_mu);
    const int64_t h = g_win_next++;
    g_windows[h] = w;
    return h;
}

int deft_window_free(int64_t window) {
    std::lock_guard<std::mutex> lk(g_win_mu);
    return g_windows.erase(window) ? DEFT_OK : DEFT_EINVAL;
}

/* One decode step's books.  `journal`: the words deft_tree_journal_take handed over ({1 = EXTEND, node, n, n slots} | {2 = RESET, node,
 * 0}); `loc`: this step's nq slots, by query row.  replan = 0: continue the window; replan = 1: start one (the plan is being rebuilt on
 * this step).  Writes the patch list of the step -- {entries, active overflow tiles of regions 0 .. 63, {region << 20 | position,
 * row mask (0 = cleared), slot | -1 - new row} ...}, what deft_window_patch reads -- and returns the number of int32 words written; -1: this step cannot be expressed (replan = 0: run
 * a replan step; replan = 1: run the step without a window), and the books are invalid until the next replan. */
int64_t deft_window_step(int64_t window, int replan, const int32_t* journal, int64_t journal_words, const int32_t* loc, int32_t* out,
                         int64_t out_cap) {
    std::shared_ptr<WindowBooks> wp = find_window(window);
    WindowBooks* w = wp.get();
    if (!w || (journal_words > 0 && !journal) || !loc || !out || out_cap < 1 + WIN_REGIONS) {
        set_error("deft_window_step: bad arguments");
        return DEFT_EINVAL;
    }
    std::lock_guard<std::mutex> lk(w->mu);
    if (!replan && !w->valid) return -1;
    ++w->step_no;
    w->touched.clear();
    if (replan) {
        for (auto& o : w->own) o.clear();
        std::fill(w->live.begin(), w->live.end(), 0);
        std::fill(w->clean.begin(), w->clean.end(), (uint8_t)0);
        std::fill(w->fill, w->fill + WIN_REGIONS, 0);
        w->prev_new.clear();
    } else {
        // the rows the last step read from k_new / v_new are in the pool now: their positions get the pool offsets
        for (const auto& p : w->prev_new) w->put(p.pos, w->mask_of(p.pos, p.node), p.slot);
    }
    w->valid = false;  // (until this step's books are complete)
    for (int64_t at = 0; at + 2 < journal_words;) {
        const int op = journal[at], node = journal[at + 1], k = journal[at + 2];
        if (node < 0 || node >= w->n_nodes || k < 0 || at + 3 + (op == 1 ? k : 0) > journal_words) return -1;
        if (op == 2) {  // RESET: the node's slots are dropped
            if (replan) {
                w->clean[(size_t)node] = 1;
            } else {
                if (!w->clean[(size_t)node]) return -1;
                const auto& o = w->own[(size_t)node];
                for (int i = 0; i < w->live[(size_t)node]; ++i) w->put(o[(size_t)i], 0, 0);  // (no row sees the position: cleared)
                w->live[(size_t)node] = 0;
            }
            at += 3;
        } else if (op == 1) {  // EXTEND: k slots join the node
            if (replan) {
                w->clean[(size_t)node] = 0;  // (they are in the static part)
            } else {
                for (int i = 0; i < k; ++i)
                    if (!w->place(node, journal[at + 3 + i])) return -1;
            }
            at += 3 + k;
        } else {
            return -1;
        }
    }
    w->prev_new.clear();
    for (int r = 0; r < w->nq; ++r) {
        const int node = w->leaf_node[(size_t)r];
        const int before = w->live[(size_t)node];
        if (!w->place(node, -1 - r)) return -1;
        // (a leaf is below itself only: one region, one position -- unless its query's rows straddle two passes)
        for (int i = before; i < w->live[(size_t)node]; ++i) w->prev_new.push_back({w->own[(size_t)node][(size_t)i], node, loc[r]});
    }
    const int64_t n = (int64_t)w->touched.size();
    constexpr int HDR = 1 + WIN_REGIONS;
    if (n > w->max_entries || HDR + 3 * n > out_cap) return -1;
    out[0] = (int32_t)n;
    for (int h = 0; h < WIN_REGIONS; ++h) out[1 + h] = (w->fill[h] + DEFT_BLOCK_LEN - 1) / DEFT_BLOCK_LEN;
    for (int64_t i = 0; i < n; ++i) {
        const int key = w->touched[(size_t)i];
        const size_t at = (size_t)(key >> 20) * (size_t)w->cap + (size_t)(key & 0xfffff);
        out[HDR + 3 * i] = key;
        out[HDR + 3 * i + 1] = w->ent_node[at];
        out[HDR + 3 * i + 2] = w->ent_val[at];
    }
    w->valid = true;
    return HDR + 3 * n;
}

}  // extern "C"
