#!/bin/bash
# Round 3 session 2: the hand-written training step on the device (parity + speed).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests/test_train_native.py tests/test_training.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -s > $OUT/pytest_train.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_train.log
grep -v "^$" $OUT/pytest_train.log | tail -40
timeout 300 python tools/archive/runs/r3_train_time.py > $OUT/train_time.log 2>&1; cat $OUT/train_time.log | tail -30
