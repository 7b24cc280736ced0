#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F='Warning\|warn(\|amdgpu.ids\|socket.cpp\|Gloo\|^$'
echo "== with-update"; python bench.py --no-cpu-baseline --with-update --steps 100 > gpurun_out/r04_withupdate.json 2> gpurun_out/r04_withupdate.err; tail -c 400 gpurun_out/r04_withupdate.json; tail -5 gpurun_out/r04_withupdate.err
echo "== dec in fill A/B (dual)"; tools/abenv.sh "" 3 "LS2FM_DEC_IN_FILL=1" "LS2FM_DEC_IN_FILL=0" 2>&1 | tee gpurun_out/r04_ab5_decfill_dual.txt
echo "== dec in fill A/B (single)"; tools/abenv.sh "--single-field" 2 "LS2FM_DEC_IN_FILL=1" "LS2FM_DEC_IN_FILL=0" 2>&1 | tee gpurun_out/r04_ab5_decfill_single.txt
echo "== targeted tests"
timeout 1500 python -m pytest tests/test_hip_stage_loops.py tests/test_hip_dist_two_rank.py tests/test_hip_fused_render.py tests/test_hip_stage.py tests/test_hip_config_shapes.py -m gpu -q -s 2>&1 | grep -v "$F" > gpurun_out/r04_targeted_5.txt
grep -n "Fatal\|^FAILED\|passed\|failed\|Segmentation" gpurun_out/r04_targeted_5.txt | head -30
