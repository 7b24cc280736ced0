#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "" 1 "" "LS2FM_SIDE_PROBE=1" "LS2FM_SIDE_PROBE=2" "LS2FM_SIDE_PROBE=6" "LS2FM_SIDE_PROBE=4"
