#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp_ge or config3 or config4 or hidden or random_shapes or multi_member or errors or min_dist or density" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -8 gpurun_out/pytest_k.log
timeout 300 python - <<'PY' > gpurun_out/ge_rows.log 2>&1
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
from flexs_amd.utils.sequence_utils import AAS
for rep in range(2):
    ps.time_score("ge", 90, AAS, 100, 8, 100_000, label="C4 ge L=90 A=20 H=100 M=8 N=1e5")
    ps.time_score("ge", 90, AAS, 100, 8, 1_000_000, reps=5)
    ps.time_score("ge", 237, AAS, 100, 8, 100_000, reps=5)
    ps.time_score("ge", 14, "UGCA", 100, 8, 1_000_000, reps=5)
PY
grep what gpurun_out/ge_rows.log | cut -c1-200
