#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_point_queries.py tests/test_hip_stage.py -m gpu -x -q 2>&1 | tail -3
bash tools/abenv.sh "" 2 "LS2FM_SIDE_FIRST=1" "LS2FM_SIDE_FIRST=0" "LS2FM_LIB=$PWD/tools/ab/lib_fill6.so" "LS2FM_LIB=$PWD/tools/ab/lib_fill7.so"
bash tools/abenv.sh "--launch graph" 1 "LS2FM_SIDE_FIRST=1" "LS2FM_SIDE_FIRST=0"
bash tools/abenv.sh "--launch eager" 1 "LS2FM_SIDE_FIRST=1" "LS2FM_SIDE_FIRST=0"
bash tools/abenv.sh "--single-field" 1 "LS2FM_SIDE_FIRST=0"
