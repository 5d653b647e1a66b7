#!/bin/bash
# Round 4, session 2: the wide resident form -- parity tests, A/B, stress.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "resident or small_call or nam_fused or mailbox" > $OUT/pytest_resident.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_resident.log
grep -v "^\.*$" $OUT/pytest_resident.log | tail -40
timeout 500 python tools/archive/runs/r4_server_wide_ab.py > $OUT/wide_ab.log 2>&1; echo "exit: $?" >> $OUT/wide_ab.log
grep -v "amdgpu.ids" $OUT/wide_ab.log
timeout 400 python tools/archive/runs/r4_server_wide_stress.py 8000 > $OUT/wide_stress.log 2>&1; echo "exit: $?" >> $OUT/wide_stress.log
grep -v "amdgpu.ids" $OUT/wide_stress.log
