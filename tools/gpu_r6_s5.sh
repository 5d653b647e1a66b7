#!/bin/bash
# Round 6 session 5: K1 quad tail -- A/B, bit identity, the CNN test files.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s5; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_quad_tail_ab.py > $OUT/quad_tail_ab.log 2>&1; echo "exit $?" >> $OUT/quad_tail_ab.log
grep -v amdgpu.ids $OUT/quad_tail_ab.log | cut -c1-330
( time timeout 1200 python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_s5.log 2>&1 ) 2>> $OUT/pytest_s5.log
grep -v "^\.*$" $OUT/pytest_s5.log | tail -30
