#!/bin/bash
# Round 2, check session: all GPU tests + the bench line with its configs block
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log | cut -c1-250
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_check.log 2>&1; echo "exit $?" >> $OUT/bench_check.log
grep '^{' $OUT/bench_check.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps')})
r=d['roofline']; print('roof',r['kernel_ms'],r['frac'],r['frac_issued'])
for k,v in d.get('configs',{}).items(): print(k, round(v['kernel_ms']*1e3,1),'us', round(v['frac'],3), round(v['frac_issued'],3))
print(d.get('end_to_end'))
"
tail -2 $OUT/bench_check.log | cut -c1-300
