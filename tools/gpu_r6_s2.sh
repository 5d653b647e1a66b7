#!/bin/bash
# Round 6 session 2: GPU suite after the distributed / error-word / in-place changes, bench (driver form), in-kernel timelines of the target launches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 -x > $OUT/pytest_gpu.log 2>&1 ) 2>> $OUT/pytest_gpu.log
grep -v "^\.*$" $OUT/pytest_gpu.log | tail -30
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time; echo "exit $?" >> $OUT/bench_driver.time
cp gpurun_out/bench_full.json $OUT/bench_full_driver.json 2>/dev/null
wc -c $OUT/bench_driver.out; cat $OUT/bench_driver.out; cat $OUT/bench_driver.time
( time make -C flexs_amd/csrc trace -j16 > $OUT/make_trace.log 2>&1 ) 2>> $OUT/make_trace.log; tail -4 $OUT/make_trace.log
FX_SET=r6 FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace.so timeout 600 python tools/trace_probe.py > $OUT/trace_r6.log 2>&1
cp gpurun_out/trace_probe_r6.json $OUT/ 2>/dev/null
tail -3 $OUT/trace_r6.log | cut -c1-600
