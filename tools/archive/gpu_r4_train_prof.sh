#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4trainprof; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/prof -o tr -- python $GRAFT_REPO_ROOT/tools/runs/r3_train_prof.py > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_stats.csv" | head -3
cat $(find $OUT -name "*kernel_stats.csv" | head -1) | head -12
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
