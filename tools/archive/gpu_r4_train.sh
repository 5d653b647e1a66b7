#!/bin/bash
# Training step experiments: parity tests, phase timeline, wall times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4train; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_train_native.py tests/test_training.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_train.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_train.log
grep -v "^\.*$" $OUT/pytest_train.log | tail -12
timeout 200 python tools/archive/runs/r3_train_trace.py > $OUT/train_trace.log 2>&1
grep -v amdgpu $OUT/train_trace.log | head -48
timeout 300 python tools/archive/runs/r4_train_time.py > $OUT/train_time.log 2>&1
grep -v amdgpu $OUT/train_time.log
