#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo\|^$" > gpurun_out/r04_gputest_2.txt
grep -n "FAILED\|passed\|failed\|per-element" gpurun_out/r04_gputest_2.txt | tail -30
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so LS2FM_SERIAL=1 timeout 300 python tools/acc_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_acc_stamps_v3.txt
tail -22 gpurun_out/r04_acc_stamps_v3.txt
echo "== dual C2"; tools/abn.sh "" 3 cur tools/ab/lib_atomic.so tools/ab/lib_nomerge.so 2>&1 | tee gpurun_out/r04_ab_scatter_dual.txt
echo "== single"; tools/abn.sh "--single-field" 2 cur tools/ab/lib_atomic.so tools/ab/lib_nomerge.so 2>&1 | tee gpurun_out/r04_ab_scatter_single.txt
echo "== C3"; tools/abn.sh "--config C3 --steps 60" 1 cur tools/ab/lib_nomerge.so 2>&1 | tee gpurun_out/r04_ab_scatter_c3.txt
