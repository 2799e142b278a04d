# quick GPU check: multipass on/off outputs vs fp64 truth, on medusa64 / tot50 / gqa workloads
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, deft_amd
from bench import Bench
from deft_amd.utils.workloads import WORKLOADS, GEOMETRY
import deft_amd.tree_attention as ta
dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["medusa64_node", "tot50_4k", "gqa_4kx32", "northstar_4kx32"]:
    w = WORKLOADS[name]
    outs = {}
    for mp in (0, 1):
        orig = ta.multipass_launch
        if mp == 0:
            ta.multipass_launch = lambda md, Hq, Hkv, D: 0
        b = Bench(w, 2, dev, seed=1)
        print(name, "max_node_queries", b.md.max_node_queries, "mp", ta.multipass_launch(b.md, b.Hq, b.Hkv, b.D), flush=True)
        deft_amd.register_tree_metadata(b.md)
        b.step_eager(); torch.cuda.synchronize()
        o = b.attn[1](b.q[1], b.k_new[1], b.v_new[1], b.meta).clone(); torch.cuda.synchronize()
        outs[mp] = o
        # timing: graph of 2 layers x 20
        b.prepare(use_graph=True)
        for _ in range(5): b.step()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): b.step()
        e1.record(); torch.cuda.synchronize()
        print(f"  mp={mp}: {e0.elapsed_time(e1)*1e3/50/b.layers:.2f} us per layer (2 layers, cache-warm)", flush=True)
        ta.multipass_launch = orig
        if mp == 1:
            # fp64 truth for a few leaves
            paths = b.forest.leaf_paths()
            kv = b.pool.kv_data[1]
            Hq, Hkv, D = b.Hq, b.Hkv, b.D
            q = b.q[1].view(b.nq, Hq, D)
            worst = 0.0
            for r in range(0, b.nq, max(1, b.nq // 6)):
                sl = torch.tensor(paths[r], device=dev)
                k = kv[sl, 0].double().repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
                v = kv[sl, 1].double().repeat_interleave(Hq // Hkv, dim=1).transpose(0, 1)
                s = torch.einsum("hd,hsd->hs", q[r].double(), k) / D ** 0.5
                ref = torch.einsum("hs,hsd->hd", torch.softmax(s, -1), v)
                worst = max(worst, (o.view(b.nq, Hq, D)[r].double() - ref).abs().max().item())
            print("  max |err| vs fp64 truth (sampled leaves):", worst, flush=True)
        del b
    print("  max |mp1 - mp0|:", (outs[1].float() - outs[0].float()).abs().max().item(), flush=True)
