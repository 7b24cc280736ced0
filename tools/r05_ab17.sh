#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "--launch graph" 3 "" "LS2FM_JOIN_EARLY=1"
bash tools/abenv.sh "--launch eager" 2 "" "LS2FM_JOIN_EARLY=1"
