#!/bin/bash
# Round 2, GPU session 3: parity (all GPU tests), timelines after the work-split / GE byte-table changes, bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace.so timeout 400 python tools/trace_probe.py > $OUT/trace_probe.log 2>&1; echo "exit $?" >> $OUT/trace_probe.log
python - <<'PY'
import json
for l in open("gpurun_out/trace_probe.log"):
    if l.startswith("{"):
        d = json.loads(l); t = d["trace"]
        print(d["what"], "| ev %.1f us span %.1f | fill %.2f | first tile %.1f | per tile p50 %.1f | done p10/p50/p90/max %s" % (
            d["event_us_per_launch"], t["span_us"], t["fill_us_p50_max"][0], t["first_tile_dur_us_p10_p50_p90_max"][1],
            t["per_tile_us_p10_p50_p90"][1], [round(x, 1) for x in t["last_tile_done_us_p10_p50_p90_max"]]), "| phases", t["first_tile_phase_ends_us_p50"], "simd", t["waves_per_simd"])
        if "mlp" in d["what"] or "N=10000" in d["what"]: print("   block0:", t["block0_waves_simd_tiles_firststart_firstend_lastend"])
PY
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?" >> $OUT/bench.log
grep '^{' $OUT/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps')}, d.get('settled'))
r=d['roofline']; print('roof',r['kernel_ms'],r['frac'],r['frac_issued'])
for k,v in d['configs'].items(): print(k, round(v['kernel_ms']*1e3,1),'us', round(v['frac'],3), round(v['frac_issued'],3), v['reps'])
for k,v in d['member_parallel'].items(): print(k, v if isinstance(v,str) else {a:v[a] for a in ('value','ms_per_step','kernel_ms_this_rank','checked')})
"
