#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r05_gputest_full.txt; cat gpurun_out/r05_gputest_full.txt
python tools/time_loops.py 2>&1 | grep -v amdgpu > gpurun_out/r05_loops_step.txt; cat gpurun_out/r05_loops_step.txt
python tools/time_stage.py 2>&1 | grep -v amdgpu > gpurun_out/r05_stage_step.txt; cat gpurun_out/r05_stage_step.txt
rocprofv3 --kernel-trace -d gpurun_out/ltl -- python tools/loop_timeline.py run BA >/dev/null 2>&1
python tools/loop_timeline.py show gpurun_out/ltl > gpurun_out/r05_ba_loop_timeline.txt; grep "iteration span" gpurun_out/r05_ba_loop_timeline.txt; rm -rf gpurun_out/ltl
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
