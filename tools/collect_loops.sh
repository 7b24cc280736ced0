#!/bin/bash
# the stage / loop part of tools/collect_r04.sh alone (loop timings, BA iteration timeline, stage step and timeline, point queries)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/time_points.py > gpurun_out/${TAG}_point_queries.txt 2>&1
python tools/time_stage.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_stage_step.txt
python tools/time_loops.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_loops_step.txt
rocprofv3 --kernel-trace -d gpurun_out/stl -- python tools/stage_timeline.py run >/dev/null 2>&1
python tools/stage_timeline.py show gpurun_out/stl > gpurun_out/${TAG}_stage_timeline.txt; rm -rf gpurun_out/stl
rocprofv3 --kernel-trace -d gpurun_out/ltl -- python tools/loop_timeline.py run BA >/dev/null 2>&1
python tools/loop_timeline.py show gpurun_out/ltl > gpurun_out/${TAG}_ba_loop_timeline.txt; rm -rf gpurun_out/ltl
cat gpurun_out/${TAG}_stage_step.txt gpurun_out/${TAG}_loops_step.txt; grep "iteration span" gpurun_out/${TAG}_ba_loop_timeline.txt
