#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_fused_render.py tests/test_hip_fused_loss.py tests/test_hip_config_shapes.py -m gpu -q -x 2>&1 | tail -2
echo "== dual"; tools/abn.sh "" 3 cur tools/ab/lib_head.so 2>&1 | tee gpurun_out/r04_ab13_dectail_dual.txt
echo "== single"; tools/abn.sh "--single-field" 2 cur tools/ab/lib_head.so 2>&1 | tee gpurun_out/r04_ab13_dectail_single.txt
for g in 1 2; do LS2FM_DIST_SINGLE=1 timeout 200 python bench.py --force-dist --no-cpu-baseline --launch graph --shard-groups $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph, shard groups $g', round(d['ms_per_step'],4), d['launch'], d['exchange']['form'][:60])"; done
