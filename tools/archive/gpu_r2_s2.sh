#!/bin/bash
# Round 2, GPU session 2: in-kernel timelines (tools/trace_probe.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "cnn_l8 or smoke or mlp_ge" > $OUT/pytest_quick.log 2>&1
tail -2 $OUT/pytest_quick.log
timeout 400 python tools/trace_probe.py > $OUT/trace_probe.log 2>&1; echo "exit $?" >> $OUT/trace_probe.log
cat $OUT/trace_probe.log | cut -c1-1500
