#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for i in 1 2 3; do timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_relu_$i.log 2>&1; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_relu_$i.log") if l.startswith("{")][-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
done
timeout 600 python tools/perf_survey.py score > gpurun_out/perf_score.log 2>&1
cp gpurun_out/perf_survey.json gpurun_out/perf_score.json
grep what gpurun_out/perf_score.log | cut -c1-190
