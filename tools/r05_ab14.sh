#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abn.sh "" 2 cur tools/ab/lib_f2.so tools/ab/lib_f3.so tools/ab/lib_a1.so tools/ab/lib_a2.so tools/ab/lib_f2a1.so
