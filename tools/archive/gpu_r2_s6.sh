#!/bin/bash
# Round 2, quick session: a subset of the GPU tests ($1 = pytest -k expression) + the timeline set $2 of tools/trace_probe.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "$1" > $OUT/pytest_quick.log 2>&1
tail -15 $OUT/pytest_quick.log | cut -c1-250
FX_SET=$2 FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace.so timeout 600 python tools/trace_probe.py > $OUT/trace_probe_$2.log 2>&1; echo "exit $?" >> $OUT/trace_probe_$2.log
python - "$OUT/trace_probe_$2.log" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); t = d["trace"]
        print(d["what"], "| ev %.1f us span %.1f | fill %.2f | first tile start %.2f dur %.1f | done p50/max %s" % (
            d["event_us_per_launch"], t["span_us"], t["fill_us_p50_max"][0], t["first_tile_start_us_p50_max"][0], t["first_tile_dur_us_p10_p50_p90_max"][1],
            [round(x, 1) for x in t["last_tile_done_us_p10_p50_p90_max"][1::2]]))
    elif "rror" in l or "exit" in l: print(l.rstrip()[:300])
PY
