#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "fused or quad or small or explorer or api or population" > gpurun_out/pytest_fused.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_fused.log
grep -v "^$" gpurun_out/pytest_fused.log | tail -12
timeout 600 python tools/archive/runs/r3_fused_mean_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/fused_mean_ab.log
