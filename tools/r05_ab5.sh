#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "" 2 "LS2FM_FILL_REVERSE=0" "LS2FM_FILL_REVERSE=1" "LS2FM_LIB=$PWD/tools/ab/lib_nt.so" "LS2FM_FILL_REVERSE=1 LS2FM_LIB=$PWD/tools/ab/lib_nt.so"
