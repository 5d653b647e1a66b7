#!/bin/bash
# Round 6 session 6: quick K1 timing (quad tail A/B script doubles as the timing harness) + the CNN bit-identity tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s6; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_quad_tail_ab.py > $OUT/k1_timing.log 2>&1; echo "exit $?" >> $OUT/k1_timing.log
grep -v amdgpu.ids $OUT/k1_timing.log | cut -c1-200
( time timeout 1200 python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "cnn or quad or baseline" > $OUT/pytest_s6.log 2>&1 ) 2>> $OUT/pytest_s6.log
grep -v "^\.*$" $OUT/pytest_s6.log | tail -8
