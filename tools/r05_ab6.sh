#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "" 2 "LS2FM_FUSED_WGRAD=1" "LS2FM_FUSED_WGRAD=0"
bash tools/abenv.sh "--launch graph" 1 "LS2FM_FUSED_WGRAD=1" "LS2FM_FUSED_WGRAD=0"
bash tools/abenv.sh "--rays 4096" 1 "LS2FM_FUSED_WGRAD=1" "LS2FM_FUSED_WGRAD=0"
