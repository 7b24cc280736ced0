#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== stage loops (new lib)"
timeout 900 python -m pytest tests/test_hip_stage_loops.py -m gpu -q -s -x 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo\|^$" > gpurun_out/r04_loops_3.txt
grep -n "Fatal\|FAILED\|passed\|failed\|full-size\|Segmentation" gpurun_out/r04_loops_3.txt | head
if grep -q "Fatal\|Segmentation" gpurun_out/r04_loops_3.txt; then
  echo "== stage loops (HEAD lib)"
  LS2FM_LIB=$PWD/tools/ab/lib_head.so timeout 900 python -m pytest tests/test_hip_stage_loops.py -m gpu -q -s -x -k "init_loop" 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo\|^$" > gpurun_out/r04_loops_3_head.txt
  grep -n "Fatal\|FAILED\|passed\|failed\|Segmentation" gpurun_out/r04_loops_3_head.txt | head
fi
echo "== full suite"
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_hip_stage_loops.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo\|^$" > gpurun_out/r04_gputest_3.txt
grep -n "FAILED\|passed\|failed\|Fatal" gpurun_out/r04_gputest_3.txt | tail -20
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so LS2FM_SERIAL=1 timeout 300 python tools/acc_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_acc_stamps_v4.txt
tail -22 gpurun_out/r04_acc_stamps_v4.txt
echo "== dual C2"; tools/abn.sh "" 2 cur tools/ab/lib_merge24.so tools/ab/lib_merge36.so tools/ab/lib_merge65.so 2>&1 | tee gpurun_out/r04_ab3_dual.txt
echo "== dual C2 mode 0"; LS2FM_SCATTER_MODE=0 tools/abn.sh "" 2 cur 2>&1 | tee gpurun_out/r04_ab3_dual_mode0.txt
echo "== single"; tools/abn.sh "--single-field" 2 cur tools/ab/lib_merge65.so 2>&1 | tee gpurun_out/r04_ab3_single.txt
echo "== single mode 0"; LS2FM_SCATTER_MODE=0 tools/abn.sh "--single-field" 2 cur 2>&1 | tee gpurun_out/r04_ab3_single_mode0.txt
echo "== C3"; tools/abn.sh "--config C3 --steps 60" 1 cur tools/ab/lib_merge65.so 2>&1 | tee gpurun_out/r04_ab3_c3.txt
echo "== C5"; tools/abn.sh "--config C5 --steps 60" 1 cur tools/ab/lib_merge65.so 2>&1 | tee gpurun_out/r04_ab3_c5.txt
