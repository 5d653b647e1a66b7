#!/bin/bash
# Round 4, session 4: adaptive resident geometry + the one-launch fit: tests, A/B, timings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s4; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "resident or small_call or mailbox" > $OUT/pytest_resident.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_resident.log
grep -v "^\.*$" $OUT/pytest_resident.log | tail -30
timeout 600 python -m pytest tests/test_train_native.py tests/test_training.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -s > $OUT/pytest_train.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_train.log
grep -v "^\.*$" $OUT/pytest_train.log | tail -30
timeout 300 python tools/archive/runs/r4_train_time.py > $OUT/train_time.log 2>&1; echo "exit: $?" >> $OUT/train_time.log
grep -v "amdgpu.ids" $OUT/train_time.log
timeout 500 python tools/archive/runs/r4_server_wide_ab.py > $OUT/wide_ab.log 2>&1; echo "exit: $?" >> $OUT/wide_ab.log
grep -v "amdgpu.ids" $OUT/wide_ab.log
