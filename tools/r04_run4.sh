#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F='Warning\|warn(\|amdgpu.ids\|socket.cpp\|Gloo\|^$'
echo "== targeted tests"
timeout 1200 python -m pytest tests/test_hip_stage_loops.py tests/test_hip_dist_two_rank.py tests/test_hip_bench_two_rank.py tests/test_hip_checkpoint.py -m gpu -q -s 2>&1 | grep -v "$F" > gpurun_out/r04_targeted_4.txt
grep -n "Fatal\|FAILED\|passed\|failed\|Segmentation\|^E  " gpurun_out/r04_targeted_4.txt | head -30
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so LS2FM_SERIAL=1 timeout 300 python tools/acc_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_acc_stamps_v5.txt
tail -20 gpurun_out/r04_acc_stamps_v5.txt
echo "== dual C2"; tools/abn.sh "" 3 cur 2>&1 | tee gpurun_out/r04_ab4_dual.txt
echo "== dual C2 mode 0"; LS2FM_SCATTER_MODE=0 tools/abn.sh "" 2 cur 2>&1 | tee gpurun_out/r04_ab4_dual_mode0.txt
echo "== single"; tools/abn.sh "--single-field" 2 cur 2>&1 | tee gpurun_out/r04_ab4_single.txt
echo "== single mode 0"; LS2FM_SCATTER_MODE=0 tools/abn.sh "--single-field" 2 cur 2>&1 | tee gpurun_out/r04_ab4_single_mode0.txt
echo "== C3/C4/C5"; for c in C3 C4 C5; do tools/abn.sh "--config $c --steps 100" 1 cur 2>&1 | tee gpurun_out/r04_ab4_$c.txt; done
echo "== with-update"; python bench.py --no-cpu-baseline --with-update --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['update_in_step'], d.get('mirror_upkeep'))"
echo "== full suite"
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_hip_stage_loops.py --deselect tests/test_hip_dist_two_rank.py --deselect tests/test_hip_bench_two_rank.py --deselect tests/test_hip_checkpoint.py 2>&1 | grep -v "$F" > gpurun_out/r04_gputest_4.txt
grep -n "FAILED\|passed\|failed\|Fatal" gpurun_out/r04_gputest_4.txt | tail -20
