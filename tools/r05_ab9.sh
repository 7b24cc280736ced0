#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py -m gpu -x -q 2>&1 | tail -3
bash tools/abenv.sh "" 2 "LS2FM_EXPLICIT_LEVELS=0" "LS2FM_EXPLICIT_LEVELS=4" "LS2FM_EXPLICIT_LEVELS=0 LS2FM_LIB=$PWD/tools/ab/lib_minw1.so" "LS2FM_EXPLICIT_LEVELS=4 LS2FM_LIB=$PWD/tools/ab/lib_minw1.so" "LS2FM_EXPLICIT_LEVELS=3" "LS2FM_EXPLICIT_LEVELS=5"
