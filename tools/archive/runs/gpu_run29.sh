#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "pair or gfp or config5 or population or errors" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -15 gpurun_out/pytest_k.log
timeout 600 python tools/perf_survey.py protein population > gpurun_out/perf_protein.log 2>&1
cp gpurun_out/perf_survey.json gpurun_out/perf_protein.json
grep "what" gpurun_out/perf_protein.log | cut -c1-230
timeout 300 python - <<'PY' > gpurun_out/pair_big.log 2>&1
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
from flexs_amd.utils.sequence_utils import AAS
ps.time_score("cnn", 237, AAS, 100, 3, 65_536, 32, 5, reps=2, label="C5 cnn L=237 A=20 M=3 N=65536 (pair form) after SEG refactor")
ps.time_score("cnn", 237, AAS, 100, 1, 16_384, 32, 5, reps=3, label="C5 cnn L=237 A=20 M=1 N=16384 (pair form) after SEG refactor")
PY
tail -3 gpurun_out/pair_big.log | cut -c1-250
