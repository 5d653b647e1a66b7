#!/bin/bash
# Round 2, quick session: subset of parity tests + timelines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "${1:-cnn_l8 or smoke or mlp_ge or hidden or position_split or ge_byte or multi_member}" > $OUT/pytest_quick.log 2>&1
tail -4 $OUT/pytest_quick.log
FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace.so timeout 400 python tools/trace_probe.py > $OUT/trace_probe.log 2>&1; echo "exit $?" >> $OUT/trace_probe.log
tail -3 $OUT/trace_probe.log | cut -c1-300
python - <<'PY'
import json
for l in open("gpurun_out/trace_probe.log"):
    if l.startswith("{"):
        d = json.loads(l); t = d["trace"]
        print(d["what"], "| ev %.1f us span %.1f | fill %.2f | first tile %.1f | per tile p50 %.1f | done p10/p50/p90/max %s" % (
            d["event_us_per_launch"], t["span_us"], t["fill_us_p50_max"][0], t["first_tile_dur_us_p10_p50_p90_max"][1],
            t["per_tile_us_p10_p50_p90"][1], [round(x, 1) for x in t["last_tile_done_us_p10_p50_p90_max"]]), "| phases", t["first_tile_phase_ends_us_p50"], "simd", t["waves_per_simd"])
        if "mlp" in d["what"]:
            for b in t["slowest_blocks"]: print("   ", b)
PY
