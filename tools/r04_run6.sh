#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== prio A/B dual"; tools/abn.sh "" 3 cur tools/ab/lib_prio1.so tools/ab/lib_prio3.so 2>&1 | tee gpurun_out/r04_ab6_prio_dual.txt
echo "== prio A/B single"; tools/abn.sh "--single-field" 2 cur tools/ab/lib_prio3.so 2>&1 | tee gpurun_out/r04_ab6_prio_single.txt
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so LS2FM_SERIAL=1 timeout 300 python tools/acc_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_acc_stamps_v6.txt
tail -20 gpurun_out/r04_acc_stamps_v6.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20 > gpurun_out/r04_timeline_dual_a.txt; rm -rf gpurun_out/tl; cat gpurun_out/r04_timeline_dual_a.txt
rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 --single-field >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20 > gpurun_out/r04_timeline_single_a.txt; rm -rf gpurun_out/tl; cat gpurun_out/r04_timeline_single_a.txt
