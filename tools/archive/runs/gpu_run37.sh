#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp or config3 or hidden or random_shapes or multi_member or errors or pair or gfp or config5 or drop_in" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -6 gpurun_out/pytest_k.log
timeout 300 python - <<'PY' > gpurun_out/mlp_rows.log 2>&1
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
from flexs_amd.utils.sequence_utils import AAS
for rep in range(2):
    ps.time_score("mlp", 14, "UGCA", 100, 1, 100_000, label="C3 mlp L=14 H=100 M=1 N=1e5")
    ps.time_score("mlp", 14, "UGCA", 100, 1, 1_000_000, reps=5)
    ps.time_score("mlp", 90, AAS, 100, 1, 100_000, reps=5)
    ps.time_score("cnn", 237, AAS, 100, 3, 65_536, 32, 5, reps=1, label="C5 cnn L=237 A=20 M=3 N=65536 (pair form)")
PY
grep what gpurun_out/mlp_rows.log | cut -c1-200
