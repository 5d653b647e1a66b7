#!/bin/bash
# Round 4, session 5: pruned production library (losing forms in the A/B build only, compressed code objects): full GPU suite,
# the A/B build's legs, the driver's bench command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s5; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
ls -la flexs_amd/*.so >> $OUT/env.log
( time timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1 ) 2>> $OUT/pytest_gpu.log
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
grep -v "^\.*$" $OUT/pytest_gpu.log | tail -30
FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_ab.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "variants_and_tails or l14_unrolled or cnn_mfma_vs_oracle or mlp_ge_vs_oracle or software_pipelined or fused_ensemble_mean" > $OUT/pytest_ab.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_ab.log
grep -v "^\.*$" $OUT/pytest_ab.log | tail -12
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "exit $?" >> $OUT/bench_driver.log
grep '^{' $OUT/bench_driver.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps')})
print(json.dumps(d['roofline'].get('per_config')))
print(json.dumps(d['config'].get('path')))
"
