#!/bin/bash
# Quick A/B session: launch variants of K1 on the bench workload + CNN parity subset.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "cnn_l8 or smoke or baseline_config1 or random_shapes" > $OUT/pytest_quick.log 2>&1
for rep in 1 2 3; do
for v in 5 7; do
  timeout 200 python bench.py --steps 200 --warmup 20 --variant $v --no-cpu-baseline > $OUT/bench_q_v${v}_$rep.log 2>&1
done; done
tail -2 $OUT/pytest_quick.log
for f in $OUT/bench_q_*.log; do echo -n "$f "; grep '^{' $f | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip()); r=d['roofline']; print('ms/step %.4f kern_ms %.4f frac %.3f'%(d['ms_per_step'],r['kernel_ms'],r['frac']))"; done
