cd $GRAFT_REPO_ROOT; O=gpurun_out/r5h; mkdir -p $O
timeout 600 python tools/replay.py --task reasoning --golden-template docmergeToT --max-gen-len 100000 --modes flatten node --pipelined --out $O/replay_docmerge.json > $O/replay_docmerge.log 2>&1; tail -4 $O/replay_docmerge.log
(timeout 500 python tools/fuzz_session.py 360 77 2>&1 | tail -3; timeout 300 python tools/fuzz_replay.py 150 77 2>&1 | tail -2) > $O/fuzz.txt; cat $O/fuzz.txt
