#!/bin/bash
# Round 3: full GPU suite + driver-form bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s11; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
grep -v "^$" $OUT/pytest_gpu.log | tail -30
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2> $OUT/bench_driver.err; echo "exit $?" >> $OUT/bench_driver.log
grep '^{' $OUT/bench_driver.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps','rccl_ranks')}, d.get('settled'))
for k,v in d.get('configs',{}).items():
    if 'kernel_ms' in v and 'frac' in v: print(k, round(v['kernel_ms']*1e3,1),'us', round(v['frac'],3), round(v['frac_issued'],3))
for k,v in d.get('end_to_end',{}).items():
    if isinstance(v, dict) and 'wall_ms' in v: print(k, round(v['wall_ms'],3), 'ms', '%.3g'%v['value'], v.get('frac_of_kernel_rate'))
print(d.get('explorer_round'))
"
tail -3 $OUT/bench_driver.err
