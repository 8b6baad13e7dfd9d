#!/bin/bash
# kernel trace of the configs[3] batch (1 024 single-unit streams): tools/prof_c4.sh OUTDIR
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$1 -o c4 -- python3 $GRAFT_REPO_ROOT/tools/coop_bench.py c4 > $GRAFT_REPO_ROOT/$1/c4.txt 2>&1
