#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s7; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "big_string or api or chunk or drop_in or pieces" > $OUT/pytest_e2e.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_e2e.log
grep -v "^$" $OUT/pytest_e2e.log | tail -8
timeout 400 python tools/archive/runs/r3_e2e_ab.py > $OUT/e2e_ab.log 2>&1; cat $OUT/e2e_ab.log | tail -50
