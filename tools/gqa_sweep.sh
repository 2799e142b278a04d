# same-box sweep of stage-1 launch knobs over the GQA workloads:  KNOB=DEFT_NP_GRIDCAP VALS="1 2 3" bash tools/wide_ab.sh
for wl in ${WLS:-gqa_4kx32 tot50_4k forest_8kx8 forest_8kx8_single}; do
  for v in ${VALS:-1 2 3}; do
    echo -n "$wl ${KNOB:-DEFT_NP_GRIDCAP}=$v: "
    env ${KNOB:-DEFT_NP_GRIDCAP}=$v timeout 300 python bench.py --workload $wl --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step/layer %.2f us  stage1 %.2f us  frac %.3f  tok/s %.0f' % (d['attention_latency_us_per_layer'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['value']))"
  done
done
