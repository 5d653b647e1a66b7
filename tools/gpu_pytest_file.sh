#!/bin/bash
# Run GPU tests of given files / -k expression: tools/gpu_pytest_file.sh "<pytest args>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest $1 -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > gpurun_out/pytest_file.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_file.log
grep -v "^$" gpurun_out/pytest_file.log | tail -30
