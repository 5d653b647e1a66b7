#!/bin/bash
# One GPU-box session: parity tests, bench variants, rocprof kernel trace.
# Usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/archive/gpu_round.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== device"; rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -6
  echo "== build"; python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')"
} > $OUT/env.log 2>&1

echo "== pytest -m gpu" > $OUT/pytest_gpu.log
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 >> $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log

for v in 1 2 3; do
  timeout 200 python bench.py --steps 100 --warmup 10 --variant $v --no-cpu-baseline > $OUT/bench_v$v.log 2>&1
done
timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench.log 2>&1

# kernel trace + stats of the same command as the bench (summary is copied into profiles/ by hand)
rm -rf $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r1 -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/rocprof.log 2>&1
find $OUT/prof -name "*stats*" | head >> $OUT/rocprof.log
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench.log | tail -2
