#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abn.sh "" 2 cur tools/ab/lib_sbilp.so tools/ab/lib_sbmc.so tools/ab/lib_sfilp.so tools/ab/lib_sfmc.so
