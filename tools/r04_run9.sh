#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/time_stage.py 2>&1 | grep -v amdgpu > gpurun_out/r04_stage_step_a.txt; cat gpurun_out/r04_stage_step_a.txt
python tools/time_loops.py 2>&1 | grep -v amdgpu > gpurun_out/r04_loops_step_a.txt; cat gpurun_out/r04_loops_step_a.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/stl -- python tools/stage_timeline.py run >/dev/null 2>&1
python tools/stage_timeline.py show gpurun_out/stl > gpurun_out/r04_stage_timeline_a.txt; rm -rf gpurun_out/stl
cat gpurun_out/r04_stage_timeline_a.txt
rocprofv3 --kernel-trace -d gpurun_out/ltl -- python tools/loop_timeline.py run BA >/dev/null 2>&1
python tools/loop_timeline.py show gpurun_out/ltl > gpurun_out/r04_ba_loop_timeline_a.txt; rm -rf gpurun_out/ltl
tail -45 gpurun_out/r04_ba_loop_timeline_a.txt
