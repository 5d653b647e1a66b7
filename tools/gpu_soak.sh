#!/bin/bash
# The GPU suite several times over on one box (flakiness check before the round ends): tools/gpu_soak.sh [passes]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/soak; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
for i in $(seq 1 ${1:-3}); do
  ( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 -x > $OUT/pytest_$i.log 2>&1 ) 2>> $OUT/pytest_$i.log
  echo "pass $i: $(grep -E 'passed|failed|error' $OUT/pytest_$i.log | tail -1)"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
