#!/bin/bash
# the whole GPU suite + smoke: tools/gpu_tests.sh [extra pytest args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 -x $1 > gpurun_out/pytest_gpu.log 2>&1 ) 2>> gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
grep -v "^\.*$" gpurun_out/pytest_gpu.log | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
