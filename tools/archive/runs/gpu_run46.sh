#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "hidden or mlp_ge or random_shapes or multi_member or config3 or errors or drop_in" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -12 gpurun_out/pytest_k.log
timeout 300 python - <<'PY' > gpurun_out/slab_rows.log 2>&1
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
from flexs_amd.utils.sequence_utils import AAS
for slab in (0, 1):
    o = {"dense_slab": slab}
    ps.time_score("mlp", 14, "UGCA", 200, 1, 100_000, label=f"mlp L=14 H=200 M=1 N=1e5 slab={slab}", opts=o)
    ps.time_score("mlp", 14, "UGCA", 200, 1, 1_000_000, reps=5, label=f"mlp L=14 H=200 M=1 N=1e6 slab={slab}", opts=o)
    ps.time_score("mlp", 14, "UGCA", 256, 1, 1_000_000, reps=5, label=f"mlp L=14 H=256 M=1 N=1e6 slab={slab}", opts=o)
    ps.time_score("mlp", 90, AAS, 200, 1, 100_000, reps=5, label=f"mlp L=90 A=20 H=200 M=1 N=1e5 slab={slab}", opts=o)
    ps.time_score("ge", 90, AAS, 200, 8, 100_000, reps=5, label=f"ge L=90 A=20 H=200 M=8 N=1e5 slab={slab}", opts=o)
    ps.time_score("ge", 90, AAS, 256, 1, 1_000_000, reps=5, label=f"ge L=90 A=20 H=256 M=1 N=1e6 slab={slab}", opts=o)
PY
grep what gpurun_out/slab_rows.log | cut -c1-200
