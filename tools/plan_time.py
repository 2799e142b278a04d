import torch, time, sys, os
sys.path.insert(0, os.getcwd())
import deft_amd
from deft_amd.memory_pool import ReqToTokenPool, TokenToKVPool
from deft_amd.tree_cache import TreeCache
def flat_args(md): return (md.block_len, md.block_q, md.block_q_cnts, md.block_q_offset, md.block_bitmasks, md.block_kv, md.block_lens)
for (Hq, Hkv, prefix, width) in [(4, 2, 100_000, 48), (2, 1, 450_000, 3), (32, 32, 4096, 32)]:
    D = 128
    size = prefix + 4 * width + 256
    req = ReqToTokenPool(width + 8, size + 8, device="cuda")
    pool = TokenToKVPool(size, torch.float16, Hkv, D, 1, device="cuda")
    tree = TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    tree.init_prompt(torch.arange(1, prefix + 1, dtype=torch.int32))
    tree.branch(tree.root, width)
    for _ in range(3):
        for leaf in list(tree.leaves.values()): leaf.append_token(7)
        tree.alloc()
    t = time.perf_counter(); md = deft_amd.TreeMetadata.from_tree_cache(tree); torch.cuda.synchronize(); t_md = time.perf_counter() - t
    q = torch.randn((width, Hq, D), dtype=torch.float16, device="cuda")
    kb, vb = pool.get_key_buffer(0), pool.get_value_buffer(0)
    o = torch.empty_like(q)
    res = {}
    for mode in ("flatten", "node"):
        def call():
            if mode == "flatten": deft_amd.tree_attention_subtree_fwd(q, kb, vb, o, *flat_args(md))
            else: deft_amd.tree_attention_fwd(q, kb, vb, o, md.node_kv, md.node_kv_offset, md.node_kv_len, md.node_q, md.node_q_offset, md.node_q_len)
        # first call builds the plan, later calls reuse it
        md2 = deft_amd.TreeMetadata.from_tree_cache(tree)  # fresh tensors -> fresh plan
        md_save, md = md, md2
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); call(); e2.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3)
        md = md_save
    print(f"Hq={Hq} Hkv={Hkv} prefix={prefix} width={width}: metadata {t_md*1e3:.1f} ms;", {k: f"first call {a:.0f} us (plan + attention), next {b:.0f} us" for k, (a, b) in res.items()})
