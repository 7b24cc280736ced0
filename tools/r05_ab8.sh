#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_fused_loss.py tests/test_hip_stage.py tests/test_hip_point_queries.py -m gpu -x -q 2>&1 | tail -4
bash tools/abenv.sh "" 2 "LS2FM_EXPLICIT_LEVELS=0" "LS2FM_EXPLICIT_LEVELS=3" "LS2FM_EXPLICIT_LEVELS=4" "LS2FM_EXPLICIT_LEVELS=5"
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so python tools/acc_stamps_p.py 2>&1 | tail -22
