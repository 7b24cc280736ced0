#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abn.sh "" 2 cur tools/ab/lib_dec64.so tools/ab/lib_dec256.so
bash tools/abenv.sh "" 1 "LS2FM_SIDE_IN_FILL=0"
bash tools/abenv.sh "--config C1" 1 ""
bash tools/abenv.sh "--single-field" 1 "LS2FM_SIDE_IN_FILL=0" ""
bash tools/timeline.sh r05c | tail -9
