#!/bin/bash
# Full parity + end-to-end / small-call latency survey.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
timeout 600 python tools/perf_survey.py e2e nam > $OUT/perf_e2e.log 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -2 $OUT/pytest_gpu.log; grep "small call\|end-to-end\|NoisyAbstract" $OUT/perf_e2e.log | cut -c1-230; grep '^{' $OUT/bench.log | cut -c1-260
