#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so python tools/acc_stamps_p.py 2>&1 | tail -25
bash tools/abenv.sh "" 1 "LS2FM_LEVEL_GROUPS=1" "LS2FM_LEVEL_GROUPS=2" "LS2FM_LEVEL_GROUPS=4"
