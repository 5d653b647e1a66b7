#!/bin/bash
# Run one python script on the GPU box: tools/gpu_run_py.sh path/to/script.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 600 python $1 > gpurun_out/run_py.log 2>&1
echo "exit: $?" >> gpurun_out/run_py.log
grep -v "amdgpu.ids" gpurun_out/run_py.log | tail -60
