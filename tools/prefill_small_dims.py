import torch, json, sys, os
sys.path.insert(0, os.getcwd())
import deft_amd
for (Hq, Hkv, D) in ((32, 32, 64), (64, 8, 64), (32, 32, 32), (32, 32, 16)):
    for S in (4096, 16384):
        if D < 64 and S > 4096: continue
        qkv = torch.randn((S, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda")
        q, k, v = (t.view(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
        o = torch.empty((S, Hq, D), dtype=torch.float16, device="cuda")
        start = torch.zeros(1, dtype=torch.int32, device="cuda"); lens = torch.tensor([S], dtype=torch.int32, device="cuda")
        for _ in range(3): deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): deft_amd.context_attention_fwd(q, k, v, o, start, lens, S)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        fl = 2.0 * 2.0 * (S * (S + 1) / 2) * D * Hq
        print(json.dumps({"Hq": Hq, "Hkv": Hkv, "D": D, "S": S, "us": round(us, 1), "TFLOPs": round(fl / us / 1e6, 1)}))
