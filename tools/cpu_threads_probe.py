import sys, time, torch
sys.path.insert(0, "/root/repo")
from oracle.cpu_baseline import sequential_attention_cpu
torch.manual_seed(0)
nq, Hq, D, S = 32, 32, 128, 4296
q = torch.randn(nq, Hq, D); kv = torch.randn(10752, 2, Hq, D)
paths = [torch.randint(0, 10752, (S,)) for _ in range(nq)]
for thr in (8, 16, 32, 64, 128):
    torch.set_num_threads(thr)
    sequential_attention_cpu(q[:4], kv, paths[:4])
    t0 = time.perf_counter(); sequential_attention_cpu(q, kv, paths); dt = time.perf_counter() - t0
    print(thr, round(dt * 1e3, 1), "ms per layer-step")
