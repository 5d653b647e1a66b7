#!/bin/bash
# population-step tests + timing, additive-sum timing, bench with the extra CPU figures
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "population or additive or decode" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -15 gpurun_out/pytest_k.log
timeout 300 python tools/perf_survey.py population > gpurun_out/perf_population.log 2>&1
cp gpurun_out/perf_survey.json gpurun_out/perf_population.json
tail -5 gpurun_out/perf_population.log
timeout 400 python bench.py --steps 200 --warmup 20 --cpu-nam > gpurun_out/bench_cpu_extra.log 2>&1
tail -1 gpurun_out/bench_cpu_extra.log | cut -c1-300
