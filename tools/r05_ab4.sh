#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abn.sh "" 1 cur tools/ab/lib_fp1.so tools/ab/lib_fp2.so tools/ab/lib_fp3.so tools/ab/lib_ap1.so
bash tools/abn.sh "--single-field" 1 cur
