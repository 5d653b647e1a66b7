#!/bin/bash
# Round 6 session 3: K1 unit-quant A/B, hostile-host tests, bench with the live PMC passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s3; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python tools/runs/r6_unit_quant_ab.py > $OUT/unit_quant_ab.log 2>&1; echo "exit $?" >> $OUT/unit_quant_ab.log
grep -v amdgpu.ids $OUT/unit_quant_ab.log | cut -c1-260
( time timeout 900 python -m pytest tests/test_gpu_hostile_host.py tests/test_gpu_launch_first.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_hostile.log 2>&1 ) 2>> $OUT/pytest_hostile.log
grep -v "^\.*$" $OUT/pytest_hostile.log | tail -40
for m in idle oversubscribed foreign; do timeout 120 python tools/runs/r6_hostile_host.py $m 4 2>/dev/null | grep '^{' >> $OUT/hostile_host.log; done
cat $OUT/hostile_host.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time; echo "exit $?" >> $OUT/bench_driver.time
cp gpurun_out/bench_full.json $OUT/bench_full_driver.json 2>/dev/null
wc -c $OUT/bench_driver.out; cut -c1-1500 $OUT/bench_driver.out; cat $OUT/bench_driver.time
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6s3/bench_full_driver.json"))
print(json.dumps(d.get("live_pmc"), indent=1)[:1500])
PY
