#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 600 python - <<'PY' > gpurun_out/big_units.log 2>&1
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
for L, alpha in ((8, "TGCA"), (14, "UGCA")):
    for M, N in ((1, 100_000), (1, 50_000), (3, 30_000), (1, 200_000), (3, 50_000)):
        for bu in (32, 24, 16, 8, 4):
            ps.time_score("cnn", L, alpha, 100, M, N, 32, 5, label=f"cnn L={L} M={M} N={N} big_units={bu}", opts={"cnn_big_units": bu})
PY
grep what gpurun_out/big_units.log | python3 -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l.strip()); print(d['what'], round(d['kernel_ms']*1e3,1),'us', round(d['frac_mfma_peak'],3))
"
