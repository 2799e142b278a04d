"""bench.py as the driver runs it (GPU box): the JSON contract of the default line, BASELINE configs[4] measured two ways in
one run, and the N > 1 path -- launcher, process group, barrier / MAX bracket, sharded forest, its advancing loop -- rehearsed
with two ranks on ONE GPU over gloo (`--dist-backend gloo`; the driver's 8-GPU run uses RCCL and one GPU per rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--no-extras", "--no-cpu-baseline", "--no-traffic", "--no-e2e"]


def _line(cmd, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_cfg5_measured_two_ways_agrees_and_contract_keys():
    """`--workload forest_8kx8` (8 trees of 8192 x 8 x 64 in one pool, one Forest call per layer) as the HEADLINE, and the
    `cfg5_sharded_forest` line of the same run (shard_trees -> this rank's 8 trees): the same launches, so the same time."""
    d = _line([sys.executable, "bench.py", "--workload", "forest_8kx8", "--steps", "80", "--warmup", "10", *FAST])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 80 and d["warmup"] == 10 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    c5 = d["cfg5_sharded_forest"]
    assert c5["trees_this_rank"] == list(range(8)) and c5["collectives_in_data_path"] == 0
    head = d["attention_latency_us_per_layer"]
    assert abs(c5["us_per_layer"] - head) / head < 0.08, (c5["us_per_layer"], head)
    assert abs(c5["tokens_per_s"] - d["value"]) / d["value"] < 0.08
    assert "error" not in (c5["end_to_end"] or {})


def test_two_ranks_over_gloo_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--steps", "20", "--warmup", "5",
               *FAST])
    assert d["n_gpus"] == 2 and d["dist"]["backend"] == "gloo" and d["dist"]["world_size"] == 2
    c5 = d["cfg5_sharded_forest"]
    a, b = c5["trees_by_rank"]
    assert len(a) == len(b) == 8 and not set(a) & set(b) and sorted(a + b) == list(range(16))
    assert c5["collectives_in_data_path"] == 0 and c5["end_to_end"] and "error" not in c5["end_to_end"]
    assert d["value"] > 0 and c5["tokens_per_s"] > 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_bracket_at_world_size_one():
    """RCCL itself (backend "nccl"), one rank on the one GPU: `init_process_group("nccl", device_id=...)`, the rehearsed barrier /
    MAX all-reduce with the control tensors ON THE GPU (`_ctl_device`), the distributed bracket around the timed region and around
    the sharded forest's loop, `destroy_process_group` -- everything of the N > 1 path except a second GPU."""
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "20",
               "--warmup", "5", *FAST])
    assert d["n_gpus"] == 1 and d["dist"]["backend"] == "nccl" and d["dist"]["world_size"] == 1 and not d["dist"]["ranks_share_gpus"]
    c5 = d["cfg5_sharded_forest"]
    assert c5["dist_backend"] == "nccl" and c5["trees_by_rank"] == [list(range(8))] and c5["collectives_in_data_path"] == 0
    assert d["value"] > 0 and "error" not in (c5["end_to_end"] or {})


@pytest.mark.timeout(1500)
def test_eight_ranks_over_gloo_on_one_gpu():
    """BASELINE configs[4] at its full rank count -- 64 trees, 8 per rank -- with all eight ranks sharing this GPU over gloo: the
    launcher, the process group, `cfg5_shard` at world size 8 and the bracket as the driver's 8-GPU run will execute them."""
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--dist-backend", "gloo", "--steps", "8", "--warmup", "2",
               *FAST], timeout=1400)
    assert d["n_gpus"] == 8 and d["dist"]["world_size"] == 8 and d["dist"]["ranks_share_gpus"]
    c5 = d["cfg5_sharded_forest"]
    shares = c5["trees_by_rank"]
    assert len(shares) == 8 and all(len(sh) == 8 for sh in shares)
    assert sorted(t for sh in shares for t in sh) == list(range(64))  # every tree once
    assert c5["collectives_in_data_path"] == 0 and "error" not in (c5["end_to_end"] or {})
    assert d["value"] > 0 and c5["tokens_per_s"] > 0
