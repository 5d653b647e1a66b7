#!/bin/bash
# Run a subset of the GPU parity tests: tools/gpu_pytest_k.sh "<pytest -k expression>"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "$1" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -40 gpurun_out/pytest_k.log
