#!/bin/bash
# Quick interleaved A/B session for K1 launch / scheduling variants on the bench workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "cnn_l8 or smoke or l14" > $OUT/pytest_quick.log 2>&1
rm -f $OUT/bench_q_*.log
for rep in 1 2 3; do
for v in 7 8 9; do
  timeout 200 python bench.py --steps 200 --warmup 20 --variant $v --no-cpu-baseline > $OUT/bench_q_v${v}_$rep.log 2>&1
done; done
timeout 300 python - > $OUT/l14_ab.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import tools.perf_survey as ps
for rep in range(3):
    for v in (6, 10):
        ps.time_score("cnn", 14, "UGCA", 100, 3, 100_000, 32, 5, variant=v, label=f"cnn L=14 M=3 N=1e5 variant {v}")
PY
tail -2 $OUT/pytest_quick.log
for f in $OUT/bench_q_*.log; do echo -n "$f "; grep '^{' $f | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip()); r=d['roofline']; print('ms/step %.4f kern_ms %.4f frac %.3f'%(d['ms_per_step'],r['kernel_ms'],r['frac']))"; done
grep "variant" $OUT/l14_ab.log | cut -c1-110
