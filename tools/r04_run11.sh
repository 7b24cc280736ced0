#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F='Warning\|warn(\|amdgpu.ids\|socket.cpp\|Gloo\|^$'
timeout 600 python -m pytest tests/test_hip_camera_rays.py -m gpu -q -s 2>&1 | grep -v "$F" | tail -25
timeout 900 python -m pytest tests/test_hip_stage_loops.py tests/test_hip_stage.py -m gpu -q -s 2>&1 | grep -v "$F" > gpurun_out/r04_loops_11.txt; grep -n "Fatal\|^FAILED\|passed\|failed\|\[ba \|^E  " gpurun_out/r04_loops_11.txt | head
python tools/time_loops.py 2>&1 | grep -v amdgpu
for g in 1 2; do timeout 180 python tools/exp_capture_dist.py $g 2>&1 | grep "n_groups"; done
for g in 1 2 4; do LS2FM_DIST_SINGLE=1 timeout 200 python bench.py --force-dist --no-cpu-baseline --shard-groups $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shard groups $g', round(d['ms_per_step'],4), d['exchange']['form'][:60])"; done
