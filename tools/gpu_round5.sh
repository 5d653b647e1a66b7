#!/bin/bash
# Round 5 validation + profile session: GPU tests (with durations), bench lines, kernel trace, PMC passes (separate runs, --pmc only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5f; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=30 > $OUT/pytest_gpu.log 2>&1 ) 2>> $OUT/pytest_gpu.log
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
grep -v "^\.*$" $OUT/pytest_gpu.log | tail -45
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "exit $?" >> $OUT/bench_driver.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; echo "exit $?" >> $OUT/bench.log
B="python bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o r5 -- $B > $OUT/rocprof.log 2>&1
P="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o r5 -- $P > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o r5 -- $P > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o r5 -- $P > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/pmc_inst -o r5 -- $P > $OUT/pmc_inst.log 2>&1
python tools/summarize_pmc.py $OUT/pmc_bench.json $OUT/pmc_bench.md fetch=$OUT/pmc_fetch write=$OUT/pmc_write sq=$OUT/pmc_sq inst=$OUT/pmc_inst > $OUT/pmc_bench_summary.log 2>&1
# (back in the repo: python tools/flatten_pmc.py gpurun_out/r5f/pmc_bench.json profiles/r5_pmc_bench.json <commit>  -- the form bench.py reads)
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_lp -o lp -- python tools/runs/r4_lp_prof.py > $OUT/rocprof_lp.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_train -o tr -- python tools/runs/r3_train_prof.py > $OUT/rocprof_train.log 2>&1
timeout 200 python tools/runs/r5_launch_first.py > $OUT/launch_first.log 2>&1
timeout 100 python tools/runs/r5_launch_first_parts.py >> $OUT/launch_first.log 2>&1
timeout 100 python tools/runs/r5_e2e_breakdown.py > $OUT/e2e_breakdown.log 2>&1
timeout 100 python tools/runs/r5_stage_host_ab.py > $OUT/stage_host_ab.log 2>&1
timeout 200 python tools/runs/r5_relay.py > $OUT/relay.log 2>&1
cp gpurun_out/parity_error_stats.json $OUT/ 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*_kernel_trace.csv" -size +1M -delete
du -sh $OUT
for f in bench_driver bench; do echo "== $f"; grep '^{' $OUT/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps')}, d.get('settled'))
r=d['roofline']; print('roof',r['kernel_ms'],r['frac'],r['frac_issued'], r.get('mfma_util_pmc'), r.get('pmc_source'))
print({k: v for k, v in r.items() if not isinstance(v, (dict, list, str))})
"; done
cat $OUT/pmc_bench.md | head -30
head -8 $(find $OUT/prof -name "*kernel_stats.csv" | head -1)
head -8 $(find $OUT/prof_lp -name "*kernel_stats.csv" | head -1)
