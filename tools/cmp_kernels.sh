for k in stream np; do
  DEFT_STAGE1_KERNEL=$k DEFT_NP_CHUNK=8 DEFT_NP_UNION=2 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/s4_$k.json 2> gpurun_out/s4_$k.err
  python - <<PY
import json
d=json.load(open("gpurun_out/s4_$k.json"))
print("$k", "main: step/layer", d["attention_latency_us_per_layer"], "stage1", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"])
for n,v in d["other_workloads"].items():
    print("   ", n, v.get("us_per_layer"), v.get("stage1_us"), v.get("stage1_hbm_frac"), v.get("error"))
PY
done
