#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_point_queries.py tests/test_hip_fused_loss.py -m gpu -x -q 2>&1 | tail -5
bash tools/abenv.sh "" 2 "LS2FM_ACC_PERSISTENT=0" "LS2FM_ACC_PERSISTENT=1"
bash tools/abenv.sh "--single-field" 1 "LS2FM_ACC_PERSISTENT=0" "LS2FM_ACC_PERSISTENT=1"
