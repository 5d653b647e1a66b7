#!/bin/bash
# Round 6 session 1: does the short contract line reach stdout the way the driver reads it; GPU suite at the round's start.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s1; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time; echo "exit $?" >> $OUT/bench_driver.time
cp gpurun_out/bench_full.json $OUT/bench_full_driver.json 2>/dev/null
wc -c $OUT/bench_driver.out; cat $OUT/bench_driver.out; cat $OUT/bench_driver.time
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=15 > $OUT/pytest_gpu.log 2>&1 ) 2>> $OUT/pytest_gpu.log
grep -v "^\.*$" $OUT/pytest_gpu.log | tail -30
