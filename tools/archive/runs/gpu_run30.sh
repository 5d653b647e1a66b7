#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "cnn or smoke or config1 or drop_in or errors or random_shapes or hidden" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -15 gpurun_out/pytest_k.log
timeout 600 python tools/perf_survey.py protein > gpurun_out/perf_protein.log 2>&1
cp gpurun_out/perf_survey.json gpurun_out/perf_small_calls.json
grep "A=4" gpurun_out/perf_protein.log | cut -c1-230
for i in 1 2; do timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_seg_$i.log 2>&1; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_seg_$i.log") if l.startswith("{")][-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
done
