#!/bin/bash
# Resident small-call form: parity test, then the A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3srv
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/r3srv/env.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -k "resident_small_call or fused_ensemble_mean or small" -m gpu -q --tb=short -p no:cacheprovider --timeout 120 -x > gpurun_out/r3srv/pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r3srv/pytest.log
grep -v "^$" gpurun_out/r3srv/pytest.log | tail -30
timeout 400 python tools/archive/runs/r3_server_ab.py > gpurun_out/r3srv/ab.log 2>&1
echo "exit: $?" >> gpurun_out/r3srv/ab.log
grep -v "amdgpu.ids" gpurun_out/r3srv/ab.log | tail -70
timeout 300 python tools/archive/runs/r3_server_stress.py > gpurun_out/r3srv/stress.log 2>&1
echo "exit: $?" >> gpurun_out/r3srv/stress.log
grep -v "amdgpu.ids" gpurun_out/r3srv/stress.log | tail -20
