#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "cnn or config1 or config4 or multi_member or mlp_ge or hidden" > $OUT/pytest_k.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_k.log
tail -4 $OUT/pytest_k.log
for i in 1 2 3; do timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_xcd_$i.log 2>&1; python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench_xcd_$i.log") if l.startswith("{")][-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
done
rm -rf $OUT/pmc_fetch $OUT/pmc_write
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python - <<'PY'
import csv, collections
for name in ("pmc_fetch","pmc_write"):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/{name}/r1_counter_collection.csv")):
        acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'score' in k[0] or 'ensemble' in k[0]: print(k, sum(v)/len(v), len(v))
PY
