#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "big_string or drop_in or errors or config1 or round_loop or ensemble" > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_k.log
tail -15 gpurun_out/pytest_k.log
timeout 300 python tools/perf_survey.py e2e > gpurun_out/perf_e2e.log 2>&1
cp gpurun_out/perf_survey.json gpurun_out/perf_e2e.json
grep "end-to-end\|marshalling" gpurun_out/perf_e2e.log | cut -c1-220
timeout 300 python - <<'PY' 2>&1 | tail -8
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
L, alpha = 8, "TGCA"
members = [bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)]
eng = _native.Engine.get()
for N in (100_000, 1_000_000):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, alpha, 1))
    nat = [m.native() for m in members]
    for chunks in (1, 2, 3, 4, 6, 8):
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); eng.score_strings(nat, seqs, L, members[0]._lut, False, True, chunks=chunks); ts.append(time.perf_counter() - t0)
        print(f"N={N} chunks={chunks}: {np.median(ts)*1e3:.3f} ms  {N/np.median(ts):.3e} seq/s", flush=True)
PY
