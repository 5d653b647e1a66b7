#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s13; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_bench_loop_events.py > $OUT/events.log 2>&1; echo "exit $?" >> $OUT/events.log
grep -v amdgpu.ids $OUT/events.log | cut -c1-250
