#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abn.sh "" 2 cur tools/ab/lib_fnt.so tools/ab/lib_ant.so
