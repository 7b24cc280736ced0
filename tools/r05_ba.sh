#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_camera_rays.py tests/test_hip_stage_loops.py tests/test_hip_stage.py tests/test_hip_dist_two_rank.py tests/test_hip_checkpoint.py tests/test_hip_training_trajectory.py -m gpu -x -q 2>&1 | tail -3
python tools/time_loops.py 2>&1 | grep -v amdgpu
