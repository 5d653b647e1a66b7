#!/bin/bash
# Round 6 session 12: fused batch mean -- bit-identity tests, A/B of the step, CNN/forms regression.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s12; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_forms.py tests/test_gpu_api.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "mean or planes or quad_tail" > $OUT/pytest.log 2>&1 ) 2>> $OUT/pytest.log
grep -v "^\.*$" $OUT/pytest.log | tail -15
timeout 900 python tools/runs/r6_fused_mean_ab.py > $OUT/fused_mean.log 2>&1; echo "exit $?" >> $OUT/fused_mean.log
grep -v amdgpu.ids $OUT/fused_mean.log | cut -c1-250
