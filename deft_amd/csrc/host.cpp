// Host-only part of libdeft_amd.so: error text and the tree-metadata builder.
//
// deft_md_build restates TreeMetadata.from_tree_cache
// (DeFT/deft/tree_decoding/tree_cache.py:618-881) as an iterative DFS over flat
// arrays with sorted-vector set algebra instead of Python sets; its output is
// bit-identical to the reference's int64 tensors (tests/test_metadata_native.py).
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace deft {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* get_error() { return g_err; }

struct Metadata {
    int64_t query_num = 0, total_kv_len = 0;
    std::vector<int64_t> node_q, node_kv, node_q_len, node_kv_len;
    std::vector<int64_t> block_q, block_q_cnts, block_bitmasks, block_kv, block_lens;
    std::vector<int64_t> leaf_ids;
};

static std::mutex g_mu;
static std::unordered_map<int64_t, std::unique_ptr<Metadata>> g_handles;
static int64_t g_next = 1;

static std::vector<int64_t> excl_scan(const std::vector<int64_t>& v) {
    std::vector<int64_t> out(v.size());
    int64_t s = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        out[i] = s;
        s += v[i];
    }
    return out;
}

}  // namespace deft

using namespace deft;

extern "C" {

const char* deft_last_error(void) { return get_error(); }

int64_t deft_md_build(int n_nodes, const int64_t* node_id, const int64_t* parent_id, const uint8_t* is_leaf,
                      const int64_t* kv_offset, const int64_t* kv_slots, int max_q_len, int block_len,
                      int max_block_len) {
    if (n_nodes <= 0 || !node_id || !parent_id || !is_leaf || !kv_offset || (!kv_slots && kv_offset[n_nodes] > 0)) {
        set_error("deft_md_build: bad arguments");
        return DEFT_EINVAL;
    }
    if (max_q_len < 1 || max_q_len > 63 || block_len < 1 || (max_block_len < 1 && max_block_len != -1)) {
        set_error("deft_md_build: bad config max_q_len=%d block_len=%d max_block_len=%d", max_q_len, block_len,
                  max_block_len);
        return DEFT_EINVAL;
    }
    // index nodes by ascending id (children in ascending id = creation order, tree_cache.py:790)
    std::vector<int> order(n_nodes);
    for (int i = 0; i < n_nodes; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return node_id[a] < node_id[b]; });
    std::unordered_map<int64_t, int> by_id;
    by_id.reserve(n_nodes * 2);
    for (int i = 0; i < n_nodes; ++i) {
        if (!by_id.emplace(node_id[i], i).second) {
            set_error("deft_md_build: duplicate node id %lld", (long long)node_id[i]);
            return DEFT_EINVAL;
        }
    }
    int root = -1;
    std::vector<std::vector<int>> children(n_nodes);
    for (int oi = 0; oi < n_nodes; ++oi) {
        const int i = order[oi];
        if (parent_id[i] < 0) {
            if (root >= 0) {
                set_error("deft_md_build: more than one root");
                return DEFT_EINVAL;
            }
            root = i;
            continue;
        }
        auto it = by_id.find(parent_id[i]);
        if (it == by_id.end()) {
            set_error("deft_md_build: node %lld has unknown parent %lld", (long long)node_id[i], (long long)parent_id[i]);
            return DEFT_EINVAL;
        }
        children[it->second].push_back(i);  // ascending id because of `order`
    }
    if (root < 0) {
        set_error("deft_md_build: no root");
        return DEFT_EINVAL;
    }

    auto md = std::make_unique<Metadata>();
    // query rows: live leaves sorted by id (tree_cache.py:650-652)
    std::vector<int> qrow(n_nodes, -1);
    for (int oi = 0; oi < n_nodes; ++oi) {
        const int i = order[oi];
        if (is_leaf[i]) {
            qrow[i] = (int)md->leaf_ids.size();
            md->leaf_ids.push_back(node_id[i]);
        }
    }
    md->query_num = (int64_t)md->leaf_ids.size();

    // DFS pre-order and, per node, the sorted query rows of the live leaves below it (node.refs)
    std::vector<int> pre;
    pre.reserve(n_nodes);
    {
        std::vector<int> stack{root};
        while (!stack.empty()) {
            const int u = stack.back();
            stack.pop_back();
            pre.push_back(u);
            for (auto it = children[u].rbegin(); it != children[u].rend(); ++it) stack.push_back(*it);
        }
    }
    if ((int)pre.size() != n_nodes) {
        set_error("deft_md_build: %d of %d nodes unreachable from the root", n_nodes - (int)pre.size(), n_nodes);
        return DEFT_EINVAL;
    }
    std::vector<std::vector<int>> refs(n_nodes);
    for (int k = n_nodes - 1; k >= 0; --k) {  // reverse pre-order = children before parents
        const int u = pre[k];
        if (qrow[u] >= 0) refs[u].push_back(qrow[u]);
        for (int ch : children[u]) refs[u].insert(refs[u].end(), refs[ch].begin(), refs[ch].end());
        std::sort(refs[u].begin(), refs[u].end());
    }

    // running block state (tree_cache.py:653-657)
    std::vector<int64_t> cur_kv;
    struct Seg {
        int node;  // -1 = padding
        int len;
    };
    std::vector<Seg> cur_segs;
    std::vector<int> cur_union;  // sorted union of query rows of the nodes in the block
    std::vector<int> tmp;

    auto pack_block = [&]() {  // tree_cache.py:661-723
        const int cur_len = (int)cur_kv.size();
        if (cur_len < block_len) {
            cur_kv.resize(block_len, -1);
            cur_segs.push_back({-1, block_len - cur_len});
        }
        const int nqs = (int)cur_union.size();
        for (int lo = 0; lo < nqs; lo += max_q_len) {
            const int hi = std::min(nqs, lo + max_q_len);
            md->block_q.insert(md->block_q.end(), cur_union.begin() + lo, cur_union.begin() + hi);
            md->block_q_cnts.push_back(hi - lo);
            md->block_kv.insert(md->block_kv.end(), cur_kv.begin(), cur_kv.end());
            md->block_lens.push_back(cur_len);
            for (const Seg& sg : cur_segs) {
                int64_t mask = 0;
                if (sg.node >= 0) {
                    // rows of this chunk are cur_union[lo..hi); both lists are sorted
                    const std::vector<int>& rq = refs[sg.node];
                    int a = lo;
                    for (int qv : rq) {
                        while (a < hi && cur_union[a] < qv) ++a;
                        if (a >= hi) break;
                        if (cur_union[a] == qv) mask |= (int64_t)1 << (a - lo);
                    }
                }
                md->block_bitmasks.insert(md->block_bitmasks.end(), sg.len, mask);
            }
        }
        cur_kv.clear();
        cur_segs.clear();
        cur_union.clear();
    };

    auto add_piece = [&](int u, const int64_t* kv, int n) {
        cur_kv.insert(cur_kv.end(), kv, kv + n);
        cur_segs.push_back({u, n});
        tmp.clear();
        std::set_union(cur_union.begin(), cur_union.end(), refs[u].begin(), refs[u].end(), std::back_inserter(tmp));
        cur_union.swap(tmp);
    };

    {  // output sizes are known up front: every node's KV once per 32-query chunk (Node) and per block chunk (Flatten)
        size_t node_kv_n = 0;
        for (int u : pre) {
            const size_t qchunks = (refs[u].size() + (size_t)max_q_len - 1) / (size_t)max_q_len;
            node_kv_n += (size_t)(kv_offset[u + 1] - kv_offset[u]) * qchunks;
        }
        md->node_kv.reserve(node_kv_n);
        md->block_kv.reserve(node_kv_n + node_kv_n / 8 + 4 * (size_t)block_len);
        md->block_bitmasks.reserve(node_kv_n + node_kv_n / 8 + 4 * (size_t)block_len);
    }
    std::vector<int64_t> kv;
    for (int u : pre) {
        if (refs[u].empty()) {
            set_error("deft_md_build: node %lld has no live leaf below it", (long long)node_id[u]);
            return DEFT_EINVAL;
        }
        kv.assign(kv_slots + kv_offset[u], kv_slots + kv_offset[u + 1]);
        // tree_cache.py:736 sorts the node's slots; a pool hands them out ascending, so most nodes already are
        // (a 100k-token prompt: the O(n) check instead of ~1.5 ms of std::sort per step)
        if (!std::is_sorted(kv.begin(), kv.end())) std::sort(kv.begin(), kv.end());
        const int n = (int)kv.size();
        if (n == 0) {
            // A ROOT without tokens is a forest's virtual root (independent trees hanging below one tree object, so that one
            // device tree / one decode session serves the batch): it contributes nothing.  Any other empty node is the
            // reference's error (range() with step 0, tree_cache.py:746-748).
            if (u == pre[0]) continue;
            set_error("deft_md_build: node %lld has no KV slot (call alloc() first)", (long long)node_id[u]);
            return DEFT_EINVAL;
        }
        md->total_kv_len += n;
        // KV-guided grouping (tree_cache.py:744-758): q chunks outer, KV chunks inner
        const int step = (max_block_len == -1) ? n : max_block_len;
        const std::vector<int>& q = refs[u];
        for (int qlo = 0; qlo < (int)q.size(); qlo += max_q_len) {
            const int qhi = std::min((int)q.size(), qlo + max_q_len);
            for (int klo = 0; klo < n; klo += step) {
                const int khi = std::min(n, klo + step);
                md->node_q.insert(md->node_q.end(), q.begin() + qlo, q.begin() + qhi);
                md->node_q_len.push_back(qhi - qlo);
                md->node_kv.insert(md->node_kv.end(), kv.begin() + klo, kv.begin() + khi);
                md->node_kv_len.push_back(khi - klo);
            }
        }
        // flattened split (tree_cache.py:763-788)
        int room = block_len - (int)cur_kv.size();
        int done = 0;
        while (done < n) {
            if (n - done < room) {
                add_piece(u, kv.data() + done, n - done);
                break;
            }
            add_piece(u, kv.data() + done, room);
            pack_block();
            done += room;
            room = block_len;
        }
    }
    if (!cur_segs.empty()) pack_block();  // tree_cache.py:797-798

    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t hnd = g_next++;
    g_handles[hnd] = std::move(md);
    return hnd;
}

static Metadata* find(int64_t handle) {
    auto it = g_handles.find(handle);
    return it == g_handles.end() ? nullptr : it->second.get();
}

int deft_md_sizes(int64_t handle, int64_t sizes[8]) {
    std::lock_guard<std::mutex> lk(g_mu);
    Metadata* md = find(handle);
    if (!md || !sizes) {
        set_error("deft_md_sizes: bad handle");
        return DEFT_EINVAL;
    }
    sizes[0] = md->query_num;
    sizes[1] = (int64_t)md->node_q_len.size();
    sizes[2] = md->total_kv_len;
    sizes[3] = (int64_t)md->node_q.size();
    sizes[4] = (int64_t)md->node_kv.size();
    sizes[5] = (int64_t)md->block_lens.size();
    sizes[6] = (int64_t)md->block_q.size();
    sizes[7] = (int64_t)md->block_kv.size();
    return DEFT_OK;
}

static void put(int64_t* dst, const std::vector<int64_t>& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int64_t));
}

int deft_md_fetch(int64_t handle, int64_t* node_q, int64_t* node_kv, int64_t* node_q_len, int64_t* node_kv_len,
                  int64_t* node_q_offset, int64_t* node_kv_offset, int64_t* block_q, int64_t* block_q_cnts,
                  int64_t* block_q_offset, int64_t* block_bitmasks, int64_t* block_kv, int64_t* block_lens,
                  int64_t* leaf_ids) {
    std::lock_guard<std::mutex> lk(g_mu);
    Metadata* md = find(handle);
    if (!md) {
        set_error("deft_md_fetch: bad handle");
        return DEFT_EINVAL;
    }
    put(node_q, md->node_q);
    put(node_kv, md->node_kv);
    put(node_q_len, md->node_q_len);
    put(node_kv_len, md->node_kv_len);
    put(node_q_offset, excl_scan(md->node_q_len));
    put(node_kv_offset, excl_scan(md->node_kv_len));
    put(block_q, md->block_q);
    put(block_q_cnts, md->block_q_cnts);
    put(block_q_offset, excl_scan(md->block_q_cnts));
    put(block_bitmasks, md->block_bitmasks);
    put(block_kv, md->block_kv);
    put(block_lens, md->block_lens);
    put(leaf_ids, md->leaf_ids);
    return DEFT_OK;
}

int deft_md_free(int64_t handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_handles.erase(handle)) {
        set_error("deft_md_free: bad handle");
        return DEFT_EINVAL;
    }
    return DEFT_OK;
}


}  // extern "C"
