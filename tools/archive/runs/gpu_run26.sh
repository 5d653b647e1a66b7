#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "min_dist or nam or density or sharded or terminal" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -15 gpurun_out/pytest_k.log
timeout 300 python tools/perf_survey.py nam > gpurun_out/perf_nam.log 2>&1
cp gpurun_out/perf_survey.json gpurun_out/perf_nam.json
grep "min_dist\|Noisy" gpurun_out/perf_nam.log | cut -c1-200
