cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT; O=gpurun_out/w6; mkdir -p $O
python -m pytest tests/test_session.py tests/test_forest_tree.py tests/test_fuzz_slices.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
for W in legacy 1 2 4 8; do
 arg="--win-tiles $W"; [ $W = legacy ] && arg="--legacy"
 python tools/replay.py --task speculative_decoding --modes flatten node --tree-size 64 --pipelined $arg > $O/sd_${W}_$rep.log 2>&1
done; done
for f in $O/*.log; do echo -n "$f "; grep -h step_kinds $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['mode'], d['step_kinds']['replan'], d['attention_us_per_step'], end='  ')
print()
"; done
