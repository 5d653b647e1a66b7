#!/bin/bash
# Round 4, session 3: wide resident form after the second request word + plane-wise collection: tests, A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s3; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "resident or small_call or mailbox" > $OUT/pytest_resident.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_resident.log
grep -v "^\.*$" $OUT/pytest_resident.log | tail -30
timeout 500 python tools/archive/runs/r4_server_wide_ab.py > $OUT/wide_ab.log 2>&1; echo "exit: $?" >> $OUT/wide_ab.log
grep -v "amdgpu.ids" $OUT/wide_ab.log
timeout 200 python tools/archive/runs/r4_server_wide_stress.py 4000 > $OUT/wide_stress.log 2>&1; echo "exit: $?" >> $OUT/wide_stress.log
grep -v "amdgpu.ids" $OUT/wide_stress.log
