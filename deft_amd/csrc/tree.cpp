// Host-only part of libdeft_amd.so: the decoding tree over the paged KV pool.
//
// The state machine behind deft_amd.TreeCache (the reference: DeFT/deft/tree_decoding/tree_cache.py:147-403, :504-516 --
// init_prompt / branch / alloc / cut / merge_nodes / reset_node_KV / add_ref / remove_ref): nodes with a parent, children in
// creation order, the pool slots of their tokens, and which nodes are live leaves.  Stated on flat data: a node keeps the
// NUMBER of live leaves below it (the reference keeps the set, `node.refs`; membership is recomputed on demand), leaves live in
// an ordered set (query row r = the r-th live leaf by id, tree_cache.py:650-652), and cutting a leaf walks up while that count
// is zero.  The Python classes are attribute views over this object.
//
// The same object lays the tree out for the GPU (deft_tree_layout_*): nodes in DFS pre-order, every node's slots
// sorted and given room to grow, the leaf sets as bit sets -- what the device-side metadata kernels (tree_plan.h) read.
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace deft {

struct TNode {
    int64_t parent = -1;
    uint8_t leaf = 0;
    uint8_t grew = 0;   // slots were appended to it while it was NOT a live leaf (speculative decoding squeezes accepted tokens
                        // into the root every step): the next GPU layout gives it room to grow, like a leaf
    int64_t nrefs = 0;  // live leaves in the subtree (this node included)
    std::vector<int64_t> kv;        // pool slots in the order they were appended
    std::vector<int64_t> children;  // ids, creation order
};

// GPU layout of one structural epoch (tree_plan.h): valid until the structure changes or a node outgrows its room.
struct Layout {
    bool valid = false;
    int slack = 0;
    std::vector<int64_t> dfs;         // node ids in DFS pre-order
    std::unordered_map<int64_t, int> index;  // id -> DFS index
    std::vector<int32_t> start, cap;  // per DFS index: first slot of the node's region, its capacity
    std::vector<int32_t> leaf_node;   // per query row: DFS index of its leaf
    int nqw = 0;                      // 64-bit words per leaf set
    int64_t total_cap = 0;
};

// Changes the GPU layout CAN absorb besides a decode step's one-slot-per-leaf: slots appended to a node that has room, and a
// node's slots dropped.  They are journalled instead of starting a new epoch; whoever owns the device copy replays the journal
// there (tree_plan.h tree_ops_kernel) before its next use.  Words: {JOP_EXTEND, DFS index, n, n slots ...} | {JOP_RESET, DFS
// index, 0}.  What the reference's speculative-decoding mock does EVERY step (branch_func_example.py:420-437: merge_nodes of the
// accepted leaves into the root, reset_node_KV of every leaf) is exactly that.
enum : int32_t { JOP_EXTEND = 1, JOP_RESET = 2 };
constexpr size_t JOURNAL_MAX = 1 << 16;  // words; a journal that would outgrow it becomes a structural change

struct Tree {
    std::unordered_map<int64_t, TNode> nodes;
    std::set<int64_t> leaves;  // live leaves, ascending id
    int64_t root = -1;
    int64_t epoch = 1;  // bumped by every change the GPU layout cannot absorb
    Layout lay;
    std::vector<int32_t> journal;  // absorbed changes the device copy has not seen yet (valid while lay.valid)
    int64_t last_ext = -1;         // position of the journal's LAST op if it is an EXTEND (the next one to the same node joins it)
    std::mutex mu;
};

static std::mutex g_tree_mu;
static std::unordered_map<int64_t, std::shared_ptr<Tree>> g_tree_map;
static int64_t g_tree_next = 1;

// (shared ownership: a call in flight keeps its tree alive across a concurrent deft_tree_free)
static std::shared_ptr<Tree> find_tree(int64_t h) {
    std::lock_guard<std::mutex> lk(g_tree_mu);
    auto it = g_tree_map.find(h);
    return it == g_tree_map.end() ? nullptr : it->second;
}

// Lookup that cannot throw across the C ABI (std::unordered_map::at would): nullptr for an unknown id.  The tree's own
// invariant -- a node's parent exists as long as the node does: nothing with children is ever erased -- makes that a
// "cannot happen" on every walk below; the walks still stop instead of aborting the process.
static TNode* node_of(Tree* t, int64_t id) {
    auto it = t->nodes.find(id);
    return it == t->nodes.end() ? nullptr : &it->second;
}
static const TNode* node_of(const Tree* t, int64_t id) {
    auto it = t->nodes.find(id);
    return it == t->nodes.end() ? nullptr : &it->second;
}
static int64_t parent_of(const Tree* t, int64_t id) {
    const TNode* n = node_of(t, id);
    return n ? n->parent : -1;
}

static void structure_changed(Tree* t) {
    ++t->epoch;
    t->lay.valid = false;
    t->journal.clear();  // (the next upload carries everything)
    t->last_ext = -1;
}

// DFS index of a node in the current layout if the layout is valid and the journal has room for `words` more, else -1
static int absorbable(Tree* t, int64_t id, size_t words) {
    if (!t->lay.valid || t->journal.size() + words > JOURNAL_MAX) return -1;
    auto li = t->lay.index.find(id);
    return li == t->lay.index.end() ? -1 : li->second;
}

static void add_refs(Tree* t, int64_t id, int64_t d) {
    for (int64_t cur = id; cur >= 0;) {
        auto it = t->nodes.find(cur);
        if (it == t->nodes.end()) break;
        it->second.nrefs += d;
        cur = it->second.parent;
    }
}

// DFS pre-order, children in ascending id (= creation order, tree_cache.py:790)
static bool dfs_order(Tree* t, std::vector<int64_t>& out) {
    out.clear();
    if (t->root < 0) return false;
    std::vector<int64_t> stack{t->root};
    while (!stack.empty()) {
        const int64_t u = stack.back();
        stack.pop_back();
        out.push_back(u);
        const TNode* n = node_of(t, u);
        if (!n) return false;
        for (auto it = n->children.rbegin(); it != n->children.rend(); ++it) stack.push_back(*it);
    }
    return out.size() == t->nodes.size();
}

static void build_layout(Tree* t, int slack) {
    Layout& L = t->lay;
    L = Layout();
    L.slack = slack;
    if (!dfs_order(t, L.dfs)) return;
    const int n = (int)L.dfs.size();
    L.start.resize(n);
    L.cap.resize(n);
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        const TNode& nd = *node_of(t, L.dfs[i]);  // (dfs_order visited it)
        L.index[L.dfs[i]] = i;
        // live leaves grow between structural changes (one slot per decode step), and nodes that have been appended to before
        const int64_t room = (int64_t)nd.kv.size() + ((nd.leaf || nd.grew) ? slack : 0);
        L.start[i] = (int32_t)off;
        L.cap[i] = (int32_t)((room + 3) / 4 * 4);
        off += L.cap[i];
    }
    L.total_cap = off;
    L.leaf_node.clear();
    for (int64_t id : t->leaves) {
        auto li = L.index.find(id);
        if (li == L.index.end()) return;  // a live leaf the DFS did not reach: no valid layout
        L.leaf_node.push_back(li->second);
    }
    L.nqw = std::max(1, ((int)t->leaves.size() + 63) / 64);
    L.valid = off <= 0x7fffffffLL;
}

}  // namespace deft

using namespace deft;

#define DEFT_TREE_OR_FAIL(t, h, what)               \
    std::shared_ptr<Tree> t##_owner = find_tree(h); \
    Tree* t = t##_owner.get();                      \
    if (!t) {                                       \
        set_error(what ": bad tree handle");        \
        return DEFT_EINVAL;                         \
    }                                               \
    std::lock_guard<std::mutex> tree_lock(t->mu)

extern "C" {

int64_t deft_tree_create(void) {
    std::lock_guard<std::mutex> lk(g_tree_mu);
    const int64_t hnd = g_tree_next++;
    g_tree_map[hnd] = std::make_shared<Tree>();
    return hnd;
}

int deft_tree_free(int64_t tree) {
    std::lock_guard<std::mutex> lk(g_tree_mu);
    if (!g_tree_map.erase(tree)) {
        set_error("deft_tree_free: bad handle");
        return DEFT_EINVAL;
    }
    return DEFT_OK;
}

int deft_tree_add_node(int64_t tree, int64_t id, int64_t parent_id) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_add_node");
    if (id < 0 || t->nodes.count(id) || (parent_id >= 0 && !t->nodes.count(parent_id)) || (parent_id < 0 && t->root >= 0)) {
        set_error("deft_tree_add_node: id %lld exists, parent %lld unknown, or a second root", (long long)id, (long long)parent_id);
        return DEFT_EINVAL;
    }
    TNode& n = t->nodes[id];
    n.parent = parent_id;
    if (parent_id >= 0) t->nodes[parent_id].children.push_back(id);
    else t->root = id;
    structure_changed(t);
    return DEFT_OK;
}

int deft_tree_remove_node(int64_t tree, int64_t id) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_remove_node");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end() || !it->second.children.empty()) {
        set_error("deft_tree_remove_node: unknown node %lld, or it has children", (long long)id);
        return DEFT_EINVAL;
    }
    if (it->second.leaf) {
        add_refs(t, id, -1);
        t->leaves.erase(id);
    }
    const int64_t par = it->second.parent;
    if (par >= 0) {
        auto& ch = t->nodes[par].children;
        ch.erase(std::remove(ch.begin(), ch.end(), id), ch.end());
    } else {
        t->root = -1;
    }
    t->nodes.erase(it);
    structure_changed(t);
    return DEFT_OK;
}

int deft_tree_set_leaf(int64_t tree, int64_t id, int is_leaf) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_set_leaf");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end()) {
        set_error("deft_tree_set_leaf: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    const uint8_t want = is_leaf ? 1 : 0;
    if (it->second.leaf != want) {
        it->second.leaf = want;
        add_refs(t, id, want ? 1 : -1);
        if (want) t->leaves.insert(id);
        else t->leaves.erase(id);
        structure_changed(t);
    }
    return DEFT_OK;
}

// TreeCache.branch (tree_cache.py:338-370): `id` stops being a leaf, `cnt` new leaves first_id .. first_id + cnt - 1 hang below it.
int deft_tree_branch(int64_t tree, int64_t id, int cnt, int64_t first_id) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_branch");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end() || !it->second.leaf || cnt < 0) {
        set_error("deft_tree_branch: node %lld is not a live leaf", (long long)id);
        return DEFT_EINVAL;
    }
    for (int i = 0; i < cnt; ++i)
        if (t->nodes.count(first_id + i)) {
            set_error("deft_tree_branch: node id %lld exists", (long long)(first_id + i));
            return DEFT_EINVAL;
        }
    it->second.leaf = 0;
    add_refs(t, id, -1);
    t->leaves.erase(id);
    for (int i = 0; i < cnt; ++i) {
        TNode& c = t->nodes[first_id + i];
        c.parent = id;
        c.leaf = 1;
        t->nodes[id].children.push_back(first_id + i);
        add_refs(t, first_id + i, 1);
        t->leaves.insert(first_id + i);
    }
    structure_changed(t);
    return DEFT_OK;
}

// TreeCache.cut (tree_cache.py:373-403): the leaf goes, and every ancestor that is left without a live leaf below it.
// deleted_ids (leaf first, then upwards) and the slots of the deleted nodes are returned; *n_ids / *n_slots always report
// how many there are (the call fails without changing anything if a buffer is too small).
int deft_tree_cut(int64_t tree, int64_t id, int64_t* deleted_ids, int cap_ids, int* n_ids, int64_t* freed_slots,
                  int64_t cap_slots, int64_t* n_slots) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_cut");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end() || !it->second.leaf || !it->second.children.empty() || !n_ids || !n_slots) {
        set_error("deft_tree_cut: node %lld is not a childless live leaf", (long long)id);
        return DEFT_EINVAL;
    }
    // dry run: which nodes go
    std::vector<int64_t> gone;
    int64_t slots = 0;
    for (int64_t cur = id, from = -1; cur >= 0;) {
        const TNode* n = node_of(t, cur);
        if (!n || n->nrefs - 1 > 0) break;  // another live leaf below this ancestor
        // An ancestor that keeps OTHER children (nodes made by new_node() that hold no live leaf) stays: the reference pops
        // it from `nodes` and leaves those children dangling (tree_cache.py:387-397) -- a tree nothing can be built from;
        // here the walk stops, so that every node that exists keeps a parent that exists.
        if (from >= 0 && n->children.size() > 1) break;
        gone.push_back(cur);
        slots += (int64_t)n->kv.size();
        from = cur;
        cur = n->parent;
    }
    *n_ids = (int)gone.size();
    *n_slots = slots;
    if ((int)gone.size() > cap_ids || slots > cap_slots || (!deleted_ids && !gone.empty()) || (!freed_slots && slots > 0)) {
        set_error("deft_tree_cut: output buffers too small (%d nodes, %lld slots)", (int)gone.size(), (long long)slots);
        return DEFT_EWORKSPACE;
    }
    add_refs(t, id, -1);
    t->leaves.erase(id);
    int64_t w = 0;
    for (size_t k = 0; k < gone.size(); ++k) {
        const int64_t cur = gone[k];
        TNode& n = *node_of(t, cur);  // (found by the dry run above, under the same lock)
        deleted_ids[k] = cur;
        std::copy(n.kv.begin(), n.kv.end(), freed_slots + w);
        w += (int64_t)n.kv.size();
        const int64_t par = n.parent;
        if (TNode* pn = par >= 0 ? node_of(t, par) : nullptr) {
            auto& ch = pn->children;
            ch.erase(std::remove(ch.begin(), ch.end(), cur), ch.end());
        } else if (par < 0) {
            t->root = -1;
        }
        t->nodes.erase(cur);
    }
    structure_changed(t);
    return DEFT_OK;
}

// TreeCache.alloc (tree_cache.py:261-283): one new slot for every live leaf, in ascending leaf id.  Absorbed by the GPU
// layout while every leaf has room (the epoch stays; the device appends the same slots itself, tree_plan.h).
int deft_tree_alloc_step(int64_t tree, int n, const int64_t* slots) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_alloc_step");
    if (n != (int)t->leaves.size() || (n > 0 && !slots)) {
        set_error("deft_tree_alloc_step: %d slots for %d live leaves", n, (int)t->leaves.size());
        return DEFT_EINVAL;
    }
    int i = 0;
    bool fits = t->lay.valid;
    for (int64_t id : t->leaves) {
        TNode* ndp = node_of(t, id);
        if (!ndp) {
            set_error("deft_tree_alloc_step: live leaf %lld is not a node", (long long)id);
            return DEFT_EINVAL;
        }
        TNode& nd = *ndp;
        nd.kv.push_back(slots[i++]);
        if (fits) {
            auto li = t->lay.index.find(id);
            if (li == t->lay.index.end() || (int64_t)nd.kv.size() > t->lay.cap[li->second]) fits = false;
        }
    }
    if (!fits) structure_changed(t);
    return DEFT_OK;
}

int deft_tree_append_slots(int64_t tree, int n, const int64_t* ids, const int64_t* slots) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_append_slots");
    for (int i = 0; i < n; ++i) {
        auto it = t->nodes.find(ids[i]);
        if (it == t->nodes.end()) {
            set_error("deft_tree_append_slots: unknown node %lld", (long long)ids[i]);
            return DEFT_EINVAL;
        }
        it->second.kv.push_back(slots[i]);
    }
    structure_changed(t);
    return DEFT_OK;
}

int deft_tree_extend_node(int64_t tree, int64_t id, int n, const int64_t* slots) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_extend_node");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end() || n < 0) {
        set_error("deft_tree_extend_node: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    TNode& nd = it->second;
    if (n == 0) return DEFT_OK;
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] > 0x7fffffffLL) {
            set_error("deft_tree_extend_node: slot %lld out of range", (long long)slots[i]);
            return DEFT_EINVAL;
        }
    const int di = absorbable(t, id, 3 + (size_t)n);
    const bool fits = di >= 0 && n <= 1024 && (int64_t)nd.kv.size() + n <= t->lay.cap[di];  // (an op of the replay kernel holds <= 1024 slots)
    nd.kv.insert(nd.kv.end(), slots, slots + n);
    if (!nd.leaf) nd.grew = 1;
    if (fits) {  // the device copy appends the same slots itself (journal): the epoch stays
        // (merge_nodes leaf after leaf is one EXTEND per leaf to the same node: they become ONE op, one pass on the device)
        if (t->last_ext >= 0 && t->journal[(size_t)t->last_ext + 1] == di && t->journal[(size_t)t->last_ext + 2] + n <= 1024) {
            t->journal[(size_t)t->last_ext + 2] += n;
        } else {
            t->last_ext = (int64_t)t->journal.size();
            t->journal.push_back(JOP_EXTEND);
            t->journal.push_back(di);
            t->journal.push_back(n);
        }
        for (int i = 0; i < n; ++i) t->journal.push_back((int32_t)slots[i]);
        return DEFT_OK;
    }
    structure_changed(t);
    return DEFT_OK;
}

int deft_tree_set_node_kv(int64_t tree, int64_t id, int n, const int64_t* slots) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_set_node_kv");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end() || n < 0) {
        set_error("deft_tree_set_node_kv: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    if (n == 0) {  // dropping a node's slots is absorbed by the layout (its room stays)
        if (it->second.kv.empty()) return DEFT_OK;
        const int di = absorbable(t, id, 3);
        it->second.kv.clear();
        if (di >= 0) {
            t->journal.push_back(JOP_RESET);
            t->journal.push_back(di);
            t->journal.push_back(0);
            t->last_ext = -1;
            return DEFT_OK;
        }
        structure_changed(t);
        return DEFT_OK;
    }
    it->second.kv.assign(slots, slots + n);
    structure_changed(t);
    return DEFT_OK;
}

int deft_tree_clear_node_kv(int64_t tree, int64_t id) { return deft_tree_set_node_kv(tree, id, 0, nullptr); }

// The slot lists of n nodes moved out in one call: out[0 .. total) = the nodes' slots, node after node in `ids` order, each
// in append order (at most `cap` are written); every list is left empty.  Returns the total (the reference's speculative-
// decoding mock resets every leaf after every step, branch_func_example.py:430-436: one call instead of three per leaf).
int64_t deft_tree_take_nodes_kv(int64_t tree, int n, const int64_t* ids, int64_t* out, int64_t cap) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_take_nodes_kv");
    if (n < 0 || (n > 0 && !ids) || cap < 0 || (cap > 0 && !out)) {
        set_error("deft_tree_take_nodes_kv: bad arguments (n=%d)", n);
        return DEFT_EINVAL;
    }
    for (int i = 0; i < n; ++i)
        if (!t->nodes.count(ids[i])) {
            set_error("deft_tree_take_nodes_kv: unknown node %lld", (long long)ids[i]);
            return DEFT_EINVAL;
        }
    // absorbed by the layout (every node keeps its room) when all of them are in it and the journal has the space
    bool absorb = t->lay.valid && t->journal.size() + 3 * (size_t)n <= JOURNAL_MAX;
    for (int i = 0; i < n && absorb; ++i) absorb = t->lay.index.count(ids[i]) != 0;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        auto& kv = t->nodes.find(ids[i])->second.kv;
        for (const int64_t s : kv) {
            if (total < cap) out[total] = s;
            ++total;
        }
        if (absorb && !kv.empty()) {
            t->journal.push_back(JOP_RESET);
            t->journal.push_back(t->lay.index.find(ids[i])->second);
            t->journal.push_back(0);
            t->last_ext = -1;
        }
        kv.clear();
    }
    if (total > 0 && !absorb) structure_changed(t);
    return total;
}

// The journal of absorbed changes since the last call (or the last upload), oldest first; it is handed over ONCE: the caller
// replays it on the device copy (deft_tree_dev_apply_ops) before that copy is advanced or read again.  Returns the number of
// words written; a journal longer than `cap` is not handed over: the call starts a new structural epoch instead (returns -5),
// whose upload carries everything.
int64_t deft_tree_journal_take(int64_t tree, int32_t* out, int64_t cap) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_journal_take");
    const int64_t n = (int64_t)t->journal.size();
    if (n == 0) return 0;
    if (!out || n > cap) {
        structure_changed(t);
        return -5;
    }
    std::copy(t->journal.begin(), t->journal.end(), out);
    t->journal.clear();
    t->last_ext = -1;
    return n;
}

int64_t deft_tree_node_len(int64_t tree, int64_t id) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_node_len");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end()) {
        set_error("deft_tree_node_len: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    return (int64_t)it->second.kv.size();
}

// out[0 .. min(cap, len)) = the node's slots in append order; returns the node's length
int64_t deft_tree_node_kv(int64_t tree, int64_t id, int64_t* out, int64_t cap) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_node_kv");
    auto it = t->nodes.find(id);
    if (it == t->nodes.end()) {
        set_error("deft_tree_node_kv: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    const int64_t n = (int64_t)it->second.kv.size();
    if (out) std::copy(it->second.kv.begin(), it->second.kv.begin() + std::min(n, cap), out);
    return n;
}

// ids of the live leaves below (or at) a node, ascending: the reference's `node.refs`; returns how many there are
int64_t deft_tree_node_refs(int64_t tree, int64_t id, int64_t* out, int64_t cap) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_node_refs");
    if (!t->nodes.count(id)) {
        set_error("deft_tree_node_refs: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    int64_t n = 0;
    for (int64_t leaf : t->leaves) {  // a leaf is below `id` iff `id` is on its path to the root
        for (int64_t cur = leaf; cur >= 0; cur = parent_of(t, cur))
            if (cur == id) {
                if (out && n < cap) out[n] = leaf;
                ++n;
                break;
            }
    }
    return n;
}

// root -> leaf slots of a leaf's path (what sequential attention over the leaf reads); returns the path length
int64_t deft_tree_path_slots(int64_t tree, int64_t id, int64_t* out, int64_t cap) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_path_slots");
    if (!t->nodes.count(id)) {
        set_error("deft_tree_path_slots: unknown node %lld", (long long)id);
        return DEFT_EINVAL;
    }
    std::vector<int64_t> chain;
    for (int64_t cur = id; cur >= 0 && node_of(t, cur); cur = parent_of(t, cur)) chain.push_back(cur);
    int64_t n = 0;
    for (auto it = chain.rbegin(); it != chain.rend(); ++it)
        for (int64_t s : node_of(t, *it)->kv) {
            if (out && n < cap) out[n] = s;
            ++n;
        }
    return n;
}

// {nodes, live leaves, total KV slots, epoch}
int deft_tree_stats(int64_t tree, int64_t stats[4]) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_stats");
    int64_t kv = 0;
    for (const auto& kvp : t->nodes) kv += (int64_t)kvp.second.kv.size();
    stats[0] = (int64_t)t->nodes.size();
    stats[1] = (int64_t)t->leaves.size();
    stats[2] = kv;
    stats[3] = t->epoch;
    return DEFT_OK;
}

int64_t deft_tree_build_md(int64_t tree, int max_q_len, int block_len, int max_block_len) {
    std::vector<int64_t> node_id, parent_id, kv_offset, kv_slots;
    std::vector<uint8_t> is_leaf;
    {
        DEFT_TREE_OR_FAIL(t, tree, "deft_tree_build_md");
        const size_t n = t->nodes.size();
        node_id.reserve(n);
        parent_id.reserve(n);
        is_leaf.reserve(n);
        kv_offset.reserve(n + 1);
        kv_offset.push_back(0);
        size_t total = 0;
        for (const auto& kvp : t->nodes) total += kvp.second.kv.size();
        kv_slots.reserve(total);
        for (const auto& kvp : t->nodes) {
            node_id.push_back(kvp.first);
            parent_id.push_back(kvp.second.parent);
            is_leaf.push_back(kvp.second.leaf);
            kv_slots.insert(kv_slots.end(), kvp.second.kv.begin(), kvp.second.kv.end());
            kv_offset.push_back((int64_t)kv_slots.size());
        }
    }
    return deft_md_build((int)node_id.size(), node_id.data(), parent_id.data(), is_leaf.data(), kv_offset.data(),
                         kv_slots.data(), max_q_len, block_len, max_block_len);
}

// ---- GPU layout (tree_plan.h) -------------------------------------------------------------------------------
// sizes = {nodes, queries (live leaves), 64-bit words per leaf set, total slot capacity, epoch}.  (Re)builds the layout when
// the structure changed since the last call; `slack` = decode steps a leaf can grow before the next re-layout.
int deft_tree_layout(int64_t tree, int slack, int64_t sizes[5]) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_layout");
    if (slack < 0 || !sizes) {
        set_error("deft_tree_layout: bad arguments");
        return DEFT_EINVAL;
    }
    if (!t->lay.valid || t->lay.slack != slack) build_layout(t, slack);
    if (!t->lay.valid) {
        set_error("deft_tree_layout: the tree has no root, unreachable nodes, or more than 2^31 slots");
        return DEFT_EINVAL;
    }
    // A caller of this function is about to fetch an upload image, which holds every change made so far, journalled or not: the
    // pending journal is swallowed HERE (see deft_tree_layout_fetch for why that ends the epoch for every other device copy), so that
    // sizes[4] already is the epoch the image will carry -- a C caller that adopts it, the documented pattern, no longer sees a
    // mismatch at its next sync (until round 5 the bump happened inside the fetch, behind the number handed out here).
    if (!t->journal.empty()) {
        ++t->epoch;
        t->journal.clear();
        t->last_ext = -1;
    }
    sizes[0] = (int64_t)t->lay.dfs.size();
    sizes[1] = (int64_t)t->lay.leaf_node.size();
    sizes[2] = t->lay.nqw;
    sizes[3] = t->lay.total_cap;
    sizes[4] = t->epoch;
    return DEFT_OK;
}

// Fill the upload image of the current layout (host memory, sizes from deft_tree_layout):
//   node_start / node_len / node_cap [nodes] int32, refs [nodes][nqw] uint64 (bit r = query row r is below the node),
//   leaf_node [queries] int32 (DFS index of the leaf), slots [total_cap] int32 (each node's slots ASCENDING -- the
//   reference sorts them, tree_cache.py:736 -- at node_start, unused room = -1)
int deft_tree_layout_fetch(int64_t tree, int32_t* node_start, int32_t* node_len, int32_t* node_cap, uint64_t* refs,
                           int32_t* leaf_node, int32_t* slots) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_layout_fetch");
    const Layout& L = t->lay;
    if (!L.valid || !node_start || !node_len || !node_cap || !refs || !leaf_node || !slots) {
        set_error("deft_tree_layout_fetch: no valid layout (call deft_tree_layout first) or null buffer");
        return DEFT_EINVAL;
    }
    const int n = (int)L.dfs.size();
    std::fill(slots, slots + L.total_cap, -1);
    std::fill(refs, refs + (size_t)n * L.nqw, 0ull);
    std::vector<int64_t> kv;
    for (int i = 0; i < n; ++i) {
        const TNode* ndp = node_of(t, L.dfs[i]);
        if (!ndp) {
            set_error("deft_tree_layout_fetch: stale layout");
            return DEFT_EINVAL;
        }
        const TNode& nd = *ndp;
        node_start[i] = L.start[i];
        node_len[i] = (int32_t)nd.kv.size();
        node_cap[i] = L.cap[i];
        kv = nd.kv;
        if (!std::is_sorted(kv.begin(), kv.end())) std::sort(kv.begin(), kv.end());
        for (size_t k = 0; k < kv.size(); ++k) slots[L.start[i] + (int64_t)k] = (int32_t)kv[k];
    }
    int r = 0;
    for (int64_t leaf : t->leaves) {  // query row r sets its bit on every node of its path
        leaf_node[r] = L.leaf_node[r];
        for (int64_t cur = leaf; cur >= 0; cur = parent_of(t, cur)) {
            auto ci = L.index.find(cur);
            if (ci == L.index.end()) break;
            refs[(size_t)ci->second * L.nqw + (r >> 6)] |= 1ull << (r & 63);
        }
        ++r;
    }
    // The image holds every change made so far, journalled or not: whoever uploads it must not replay the journal on top of
    // it (a second device copy made inside one epoch -- another max_q_len / BLOCK_CONFIG / device -- would otherwise see the
    // pending EXTENDs twice).  Any OTHER device copy of this epoch still lacks those changes and can no longer get them from
    // the journal: a non-empty journal therefore ends the epoch for everyone (the layout itself stays valid -- the caller
    // of this function adopts the new epoch number, every other copy sees a mismatch and uploads).
    // (deft_tree_layout has already done this for the journal it found; what is caught here are changes journalled BETWEEN the two calls.)
    if (!t->journal.empty()) ++t->epoch;
    t->journal.clear();
    t->last_ext = -1;
    return DEFT_OK;
}

// Sizes of the metadata the device kernels will produce for the CURRENT lengths, each leaf `grow` tokens longer
// (0 = now; the layout's slack = the most the buffers must ever hold): sizes[0..7] as deft_md_sizes, sizes[8] = physical
// 128-slot blocks.  O(nodes + blocks), touches no slot.
// Sizes of the metadata for the current lengths + `grow` tokens per live leaf.  The leaf sets (bit sets from the leaves' paths)
// do not depend on the growth and are built once by the caller.
static void md_sizes_for(const Tree* t, const Layout& L, const std::vector<uint64_t>& refs, int max_q_len, int block_len,
                         int max_block_len, int grow, int64_t sizes[9]) {
    const int n = (int)L.dfs.size();
    std::vector<int64_t> len(n), nq(n);
    for (int i = 0; i < n; ++i) {
        const TNode& nd = *node_of(t, L.dfs[i]);  // (a valid layout names existing nodes: every erase invalidates it)
        len[i] = (int64_t)nd.kv.size() + (nd.leaf ? grow : 0);
        nq[i] = nd.nrefs;
    }
    int64_t NE = 0, total = 0, n_node_q = 0, n_node_kv = 0;
    for (int i = 0; i < n; ++i) {
        total += len[i];
        const int64_t qch = (nq[i] + max_q_len - 1) / max_q_len;
        const int64_t step = max_block_len == -1 ? len[i] : max_block_len;
        const int64_t kch = step > 0 ? (len[i] + step - 1) / step : 0;
        NE += qch * kch;
        n_node_q += nq[i] * kch;
        n_node_kv += len[i] * qch;
    }
    // flattened split: union of the leaf sets of the nodes that touch each physical block
    const int64_t NBp = (total + block_len - 1) / block_len;
    int64_t NB = 0, P = 0;
    {
        const int nqw = L.nqw;
        std::vector<uint64_t> uni(nqw);
        int i = 0;
        int64_t pos = 0;  // flattened position of node i's first slot
        for (int64_t b = 0; b < NBp; ++b) {
            const int64_t lo = b * block_len, hi = std::min(total, lo + block_len);
            std::fill(uni.begin(), uni.end(), 0ull);
            while (i < n && pos + len[i] <= lo) pos += len[i++];
            int j = i;
            int64_t pj = pos;
            while (j < n && pj < hi) {
                if (len[j] > 0)
                    for (int w = 0; w < nqw; ++w) uni[w] |= refs[(size_t)j * nqw + w];
                pj += len[j++];
            }
            int64_t c = 0;
            for (int w = 0; w < nqw; ++w) c += __builtin_popcountll(uni[w]);
            NB += (c + max_q_len - 1) / max_q_len;
            P += c;
        }
    }
    sizes[0] = (int64_t)t->leaves.size();
    sizes[1] = NE;
    sizes[2] = total;
    sizes[3] = n_node_q;
    sizes[4] = n_node_kv;
    sizes[5] = NB;
    sizes[6] = P;
    sizes[7] = NB * block_len;
    sizes[8] = NBp;
}

static std::vector<uint64_t> leaf_bitsets(const Tree* t, const Layout& L) {
    // leaf sets as sorted row lists per node would cost O(nodes x leaves); bit sets built from the leaves' paths instead
    const int n = (int)L.dfs.size(), nqw = L.nqw;
    std::vector<uint64_t> refs((size_t)n * nqw, 0ull);
    int r = 0;
    for (int64_t leaf : t->leaves) {
        for (int64_t cur = leaf; cur >= 0; cur = parent_of(t, cur)) {
            auto ci = L.index.find(cur);
            if (ci == L.index.end()) break;
            refs[(size_t)ci->second * nqw + (r >> 6)] |= 1ull << (r & 63);
        }
        ++r;
    }
    return refs;
}

// What the host builder (deft_md_build, host.cpp) and the reference refuse, the device path refuses with the same words: a
// node without a live leaf below it, and a node other than the root without a KV slot (`from_tree_cache` right after
// `branch()` with no `alloc()`: range() with step 0 upstream, tree_cache.py:746-748).  The device kernels would skip such
// nodes silently.
static int check_buildable(const Tree* t, const Layout& L, const char* who) {
    for (size_t i = 0; i < L.dfs.size(); ++i) {
        const TNode& nd = *node_of(t, L.dfs[i]);
        if (nd.nrefs <= 0) {
            set_error("%s: node %lld has no live leaf below it", who, (long long)L.dfs[i]);
            return DEFT_EINVAL;
        }
        if (nd.kv.empty() && i != 0) {
            set_error("%s: node %lld has no KV slot (call alloc() first)", who, (long long)L.dfs[i]);
            return DEFT_EINVAL;
        }
    }
    return DEFT_OK;
}

int deft_tree_md_sizes(int64_t tree, int max_q_len, int block_len, int max_block_len, int grow, int64_t sizes[9]) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_md_sizes");
    const Layout& L = t->lay;
    if (!L.valid || !sizes || max_q_len < 1 || block_len < 1 || grow < 0) {
        set_error("deft_tree_md_sizes: no valid layout or bad arguments");
        return DEFT_EINVAL;
    }
    if (grow == 0)
        if (const int rc = check_buildable(t, L, "deft_tree_md_sizes")) return rc;
    md_sizes_for(t, L, leaf_bitsets(t, L), max_q_len, block_len, max_block_len, grow, sizes);
    return DEFT_OK;
}

// Element-wise MAXIMUM of the sizes over every growth 0 .. grow_max: what buffers must hold for a whole structural epoch.
// The block arrays are NOT monotone in the leaves' lengths -- as the leaves grow the 128-slot block boundaries move over the
// nodes, and a boundary that puts pieces of several nodes into one block makes that block's query list (and, beyond 32
// queries, the number of emitted blocks) larger than it is for longer leaves: sizing for the longest tree alone
// under-allocated block_q by a few entries on multi-level trees (found by tools/fuzz_replay.py).
int deft_tree_md_sizes_upto(int64_t tree, int max_q_len, int block_len, int max_block_len, int grow_max, int64_t sizes[9]) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_md_sizes_upto");
    const Layout& L = t->lay;
    if (!L.valid || !sizes || max_q_len < 1 || block_len < 1 || grow_max < 0) {
        set_error("deft_tree_md_sizes_upto: no valid layout or bad arguments");
        return DEFT_EINVAL;
    }
    const std::vector<uint64_t> refs = leaf_bitsets(t, L);
    int64_t cur[9];
    for (int k = 0; k < 9; ++k) sizes[k] = 0;
    for (int g = 0; g <= grow_max; ++g) {
        md_sizes_for(t, L, refs, max_q_len, block_len, max_block_len, g, cur);
        for (int k = 0; k < 9; ++k) sizes[k] = std::max(sizes[k], cur[k]);
    }
    return DEFT_OK;
}

// Upper bounds of deft_tree_md_sizes over every growth 0..grow_max, in O(nodes): what an epoch's buffers are sized with.
// The node arrays, the total and the physical block count grow with the leaves, so their largest value is the one at
// grow_max.  The block arrays are not monotone (block boundaries move over the nodes as the leaves in front of them grow), and
// the exact maximum (deft_tree_md_sizes_upto) costs one pass over all blocks per growth -- 0.3 ms for a 64-leaf tree, paid
// at EVERY step of a speculative-decoding loop, where each step is a new epoch.  Bound instead: a block's query list is at
// most the concatenation of the lists of the nodes that touch it, and a node of len > 0 positions touches at most
// ceil((len - 1) / block_len) + 1 blocks wherever it starts -- exactly (last / block_len - first / block_len + 1) when
// nothing in front of it grows (the shared prefix: the nodes before the first leaf in DFS order).
int deft_tree_md_caps(int64_t tree, int max_q_len, int block_len, int max_block_len, int grow_max, int64_t sizes[9]) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_md_caps");
    const Layout& L = t->lay;
    if (!L.valid || !sizes || max_q_len < 1 || block_len < 1 || grow_max < 0) {
        set_error("deft_tree_md_caps: no valid layout or bad arguments");
        return DEFT_EINVAL;
    }
    if (const int rc = check_buildable(t, L, "deft_tree_md_caps")) return rc;
    const int n = (int)L.dfs.size();
    int64_t NE = 0, total = 0, n_node_q = 0, n_node_kv = 0, NB = 0, P = 0, fixed_pos = 0;
    bool fixed = true;  // no leaf so far: this node's first position does not move during the epoch
    for (int i = 0; i < n; ++i) {
        const TNode& nd = *node_of(t, L.dfs[i]);
        const bool grows = nd.leaf || nd.grew;  // (as build_layout gives room)
        const int64_t len = (int64_t)nd.kv.size() + (grows ? grow_max : 0), nq = nd.nrefs;
        total += len;
        const int64_t qch = (nq + max_q_len - 1) / max_q_len;
        const int64_t step = max_block_len == -1 ? len : max_block_len;
        const int64_t kch = step > 0 ? (len + step - 1) / step : 0;
        NE += qch * kch;
        n_node_q += nq * kch;
        n_node_kv += len * qch;
        if (grows && grow_max > 0) fixed = false;
        int64_t touched = 0;
        if (len > 0)
            touched = fixed ? (fixed_pos + len - 1) / block_len - fixed_pos / block_len + 1 : (len - 1 + block_len - 1) / block_len + 1;
        if (fixed) fixed_pos += len;
        NB += qch * touched;
        P += nq * touched;
    }
    // (many short nodes per block: a list holds no leaf twice, and a block of c queries is ceil(c / max_q_len) <= 1 + c / max_q_len blocks)
    const int64_t NBp = (total + block_len - 1) / block_len, nleaves = (int64_t)t->leaves.size();
    P = std::min(P, NBp * nleaves);
    NB = std::min(NB, NBp + P / max_q_len);
    sizes[0] = nleaves;
    sizes[1] = NE;
    sizes[2] = total;
    sizes[3] = n_node_q;
    sizes[4] = n_node_kv;
    sizes[5] = NB;
    sizes[6] = P;
    sizes[7] = NB * block_len;
    sizes[8] = NBp;
    return DEFT_OK;
}

// ids of the live leaves in query-row order
int deft_tree_leaf_ids(int64_t tree, int64_t* out, int cap) {
    DEFT_TREE_OR_FAIL(t, tree, "deft_tree_leaf_ids");
    int n = 0;
    for (int64_t id : t->leaves) {
        if (out && n < cap) out[n] = id;
        ++n;
    }
    return n;
}

}  // extern "C"
