#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s3; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_train_native.py tests/test_training.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_train.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_train.log
grep -v "^$" $OUT/pytest_train.log | tail -12
timeout 200 python tools/archive/runs/r3_train_trace.py > $OUT/train_trace.log 2>&1; cat $OUT/train_trace.log | head -32; timeout 300 python tools/archive/runs/r3_train_time.py 2>&1 | grep native
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/prof -o tr -- python $GRAFT_REPO_ROOT/tools/runs/r3_train_prof.py > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; head -8 $OUT/prof/*/*kernel_stats.csv 2>/dev/null || find $OUT/prof -name "*stats*" | head
