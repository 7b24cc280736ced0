#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 1 2 4; do timeout 180 python tools/exp_capture_dist.py $g 2>&1 | grep "n_groups"; done
for g in 1 2; do LS2FM_DIST_SINGLE=1 timeout 200 python bench.py --force-dist --no-cpu-baseline --launch graph --shard-groups $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph, shard groups $g', round(d['ms_per_step'],4), d['launch'], d['exchange']['form'][:60])"; done
timeout 600 python -m pytest tests/test_hip_dist_two_rank.py tests/test_hip_bench_two_rank.py -m gpu -q 2>&1 | tail -3
