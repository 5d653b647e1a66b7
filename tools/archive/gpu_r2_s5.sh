#!/bin/bash
# Round 2: first-tile phase timeline with the -DFX_TRACE_PHASES build (libflexs_amd_trace.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace_phases.so FX_PHASES=1 timeout 300 python tools/trace_probe.py > $OUT/trace_phases.log 2>&1; echo "exit $?" >> $OUT/trace_phases.log
python - <<'PY'
import json
for l in open("gpurun_out/trace_phases.log"):
    if l.startswith("{"):
        d = json.loads(l); t = d["trace"]
        print(d["what"], "| ev %.1f span %.1f | fill %.2f | first tile %.1f | phases (from tile start) %s" % (
            d["event_us_per_launch"], t["span_us"], t["fill_us_p50_max"][0], t["first_tile_dur_us_p10_p50_p90_max"][1], t["first_tile_phase_ends_us_p50"]))
PY
tail -3 $OUT/trace_phases.log | cut -c1-200
