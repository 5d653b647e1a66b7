#!/bin/bash
# Round 3 session 1: GPU tests (incl. the new multi-rank file), the driver-form bench line with the new legs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s1; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2> $OUT/bench_driver.err; echo "exit $?" >> $OUT/bench_driver.log
timeout 200 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_gpus2.log 2>&1; echo "exit $?" >> $OUT/bench_gpus2.log
tail -5 $OUT/bench_gpus2.log
grep '^{' $OUT/bench_driver.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps','rccl_ranks')}, d.get('settled'))
for k,v in d.get('configs',{}).items():
    if 'kernel_ms' in v and 'frac' in v: print(k, round(v['kernel_ms']*1e3,1),'us', round(v['frac'],3), round(v['frac_issued'],3))
    else: print(k, json.dumps(v)[:1500])
print(json.dumps(d.get('end_to_end'), indent=0)[:3000])
print(d.get('explorer_round'))
print(json.dumps(d.get('cpu_baseline'))[:1500])
"
tail -5 $OUT/bench_driver.err
