#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "generic or odd_shapes or other_kernel_sizes or random_shapes or mean_only or hidden or mlp_ge" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
grep -E "passed|failed" gpurun_out/pytest_k.log
timeout 300 python - <<'PY' 2>&1 | grep what | cut -c1-170
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
from flexs_amd.utils.sequence_utils import AAS
ps.time_score("cnn", 8, "TGCA", 100, 3, 100_000, 32, 5, reps=2, generic=True)
ps.time_score("mlp", 14, "UGCA", 100, 1, 100_000, reps=2, generic=True)
ps.time_score("ge", 90, AAS, 100, 8, 100_000, reps=2, generic=True)
ps.time_score("cnn", 14, "UGCA", 100, 1, 100_000, 64, 3, reps=2, label="cnn L=14 num_filters=64 kernel_size=3 (shape-agnostic path)")
PY
