#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "split or odd_shapes or other_kernel_sizes or random_shapes or cnn or hidden or drop_in or reference_scenarios" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
grep -E "passed|failed|^E |Error" gpurun_out/pytest_k.log | head -20
timeout 300 python - <<'PY' 2>&1 | grep what | cut -c1-200
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
ps.time_score("cnn", 14, "UGCA", 50, 3, 100_000, 32, 3, label="cnn L=14 kernel_size=3 hidden=50 M=3 N=1e5 (conv + head kernels)")
ps.time_score("cnn", 14, "UGCA", 200, 3, 100_000, 32, 4, label="cnn L=14 kernel_size=4 hidden=200 M=3 N=1e5 (conv + head kernels)")
ps.time_score("cnn", 14, "UGCA", 100, 3, 100_000, 32, 6, label="cnn L=14 kernel_size=6 hidden=100 M=3 N=1e5 (conv + head kernels)")
ps.time_score("cnn", 14, "UGCA", 100, 3, 100_000, 32, 6, reps=2, generic=True, label="same, shape-agnostic kernels")
PY
