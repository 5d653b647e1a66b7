#!/bin/bash
# Round 6 validation + profile session: GPU tests (with durations), bench lines, kernel trace, PMC passes (separate runs, --pmc only),
# the target kernels behind configs[2] / [3] / [4] and the wide hidden layers with inputs in HBM.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6f; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=20 > $OUT/pytest_gpu.log 2>&1 ) 2>> $OUT/pytest_gpu.log
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
grep -v "^\.*$" $OUT/pytest_gpu.log | tail -35
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver.time; echo "exit $?" >> $OUT/bench_driver.time
cp gpurun_out/bench_full.json $OUT/bench_full_driver.json
( time timeout 600 python bench.py > $OUT/bench.out 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "exit $?" >> $OUT/bench.time
cp gpurun_out/bench_full.json $OUT/bench_full.json
wc -c $OUT/bench_driver.out $OUT/bench.out; cat $OUT/bench_driver.out; cat $OUT/bench_driver.time
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o r6 -- $B > $OUT/rocprof.log 2>&1
P="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-live-pmc --no-settled"
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o r6 -- $P > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o r6 -- $P > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o r6 -- $P > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/pmc_inst -o r6 -- $P > $OUT/pmc_inst.log 2>&1
python tools/summarize_pmc.py $OUT/pmc_bench.json $OUT/pmc_bench.md fetch=$OUT/pmc_fetch write=$OUT/pmc_write sq=$OUT/pmc_sq inst=$OUT/pmc_inst > $OUT/pmc_bench_summary.log 2>&1
# the target kernels (inputs in HBM, launches from score_planes_dev): kernel trace, then the four counter passes
T="python tools/pmc_targets.py"
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_targets -o t -- $T > $OUT/rocprof_targets.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmcT_fetch -o t -- $T > $OUT/pmcT_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmcT_write -o t -- $T > $OUT/pmcT_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmcT_sq -o t -- $T > $OUT/pmcT_sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/pmcT_inst -o t -- $T > $OUT/pmcT_inst.log 2>&1
python tools/summarize_pmc.py $OUT/pmc_targets.json $OUT/pmc_targets.md fetch=$OUT/pmcT_fetch write=$OUT/pmcT_write sq=$OUT/pmcT_sq inst=$OUT/pmcT_inst > $OUT/pmc_targets_summary.log 2>&1
# training (configs[4]'s retrain, L = 237) kernel trace
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_train -o tr -- python tools/runs/r5_train_gfp_prof.py > $OUT/rocprof_train.log 2>&1
cp gpurun_out/parity_error_stats.json $OUT/ 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*_kernel_trace.csv" -size +1M -delete
find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT
cat $OUT/pmc_bench.md | head -12
cat $OUT/pmc_targets.md | head -40
head -12 $(find $OUT/prof -name "*kernel_stats.csv" | head -1)
head -30 $(find $OUT/prof_targets -name "*kernel_stats.csv" | head -1)
