# usage: tools/cmp_kernels.sh "VAR=a VAR2=b" "VAR=c" ...   (one bench.py run per argument, environment = the argument)
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python bench.py --no-cpu-baseline --steps 100 > gpurun_out/cmp_$i.json 2> gpurun_out/cmp_$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/cmp_$i.json"))
print("[$cfg]", "main: step/layer", d["attention_latency_us_per_layer"], "stage1", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"])
for n,v in d["other_workloads"].items():
    print("   ", n, v.get("us_per_layer"), v.get("stage1_us"), v.get("stage1_hbm_frac"), v.get("error"))
PY
done
