#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abn.sh "" 2 cur tools/ab/lib_md8.so tools/ab/lib_md16.so tools/ab/lib_md32.so
LS2FM_LIB=$PWD/tools/ab/lib_md8.so timeout 600 python -m pytest tests/test_hip_fused_render.py -m gpu -x -q 2>&1 | tail -2
