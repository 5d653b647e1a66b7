#!/bin/bash
# Interleaved A/B of the dense-kernel tile prefetch + parity of the MLP / GE paths.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp or hidden or config3 or multi_member or random_shapes" > $OUT/pytest_dense.log 2>&1
timeout 400 python - > $OUT/dense_ab.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import tools.perf_survey as ps
AAS = ps.AAS
for rep in range(3):
    for pf in (0, 1):
        o = {"dense_prefetch": pf}
        # note: time_score resets every option in `opts` to 0 afterwards; each call sets it explicitly
        ps.time_score("mlp", 14, "UGCA", 100, 1, 100_000, label=f"mlp L=14 N=1e5 prefetch={pf}", opts=o)
        ps.time_score("mlp", 14, "UGCA", 100, 1, 1_000_000, reps=5, label=f"mlp L=14 N=1e6 prefetch={pf}", opts=o)
        ps.time_score("ge", 90, AAS, 100, 8, 100_000, label=f"ge L=90 M=8 N=1e5 prefetch={pf}", opts=o)
        ps.time_score("ge", 90, AAS, 100, 8, 1_000_000, reps=5, label=f"ge L=90 M=8 N=1e6 prefetch={pf}", opts=o)
        ps.time_score("ge", 90, AAS, 100, 1, 100_000, label=f"ge L=90 M=1 N=1e5 prefetch={pf}", opts=o)
ps.eng.set_option("dense_prefetch", 1)
PY
tail -2 $OUT/pytest_dense.log; grep "prefetch=" $OUT/dense_ab.log | cut -c10-90
