#!/bin/bash
# Round 6 session 14: bench line with sampled events + the fx_score_mean_planes_dev tests (production library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s14; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_forms.py tests/test_gpu_api.py tests/test_gpu_multirank.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "mean or planes or multirank or rank" > $OUT/pytest.log 2>&1 ) 2>> $OUT/pytest.log
grep -v "^\.*$" $OUT/pytest.log | tail -8
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time; echo "exit $?" >> $OUT/bench_driver.time
cp gpurun_out/bench_full.json $OUT/bench_full_driver.json
wc -c $OUT/bench_driver.out; cat $OUT/bench_driver.out; cat $OUT/bench_driver.time | tail -4
