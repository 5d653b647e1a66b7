#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3coop
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/r3coop/env.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -k "shared_last_tiles or pipelined or mlp_small or dense" -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x > gpurun_out/r3coop/pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r3coop/pytest.log
grep -v "^$" gpurun_out/r3coop/pytest.log | tail -15
timeout 400 python tools/archive/runs/r3_dense_coop_ab.py > gpurun_out/r3coop/ab.log 2>&1
echo "exit: $?" >> gpurun_out/r3coop/ab.log
grep -v "amdgpu.ids" gpurun_out/r3coop/ab.log | tail -60
