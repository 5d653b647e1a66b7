#!/bin/bash
# Round 3 validation + profile session: GPU tests, bench lines, kernel trace, PMC passes (separate runs, --pmc only), survey.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3d; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "exit $?" >> $OUT/bench_driver.log
timeout 300 python bench.py > $OUT/bench.log 2>&1; echo "exit $?" >> $OUT/bench.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --no-cpu-baseline > $OUT/bench_dist1.log 2>&1; echo "exit $?" >> $OUT/bench_dist1.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --mode member --no-cpu-baseline --no-extras > $OUT/bench_dist1_member.log 2>&1; echo "exit $?" >> $OUT/bench_dist1_member.log
B="python bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o r3 -- $B > $OUT/rocprof.log 2>&1
P="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o r3 -- $P > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o r3 -- $P > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o r3 -- $P > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/pmc_inst -o r3 -- $P > $OUT/pmc_inst.log 2>&1
python tools/summarize_pmc.py $OUT/pmc_bench.json $OUT/pmc_bench.md fetch=$OUT/pmc_fetch write=$OUT/pmc_write sq=$OUT/pmc_sq inst=$OUT/pmc_inst > $OUT/pmc_bench_summary.log 2>&1
T="python tools/pmc_targets.py"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmcT_sq -o t -- $T > $OUT/pmcT_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/pmcT_inst -o t -- $T > $OUT/pmcT_inst.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES -f csv -d $OUT/pmcT_act -o t -- $T > $OUT/pmcT_act.log 2>&1
python tools/summarize_pmc.py $OUT/pmc_targets.json $OUT/pmc_targets.md sq=$OUT/pmcT_sq inst=$OUT/pmcT_inst act=$OUT/pmcT_act > $OUT/pmc_targets_summary.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/profT -o t -- $T > $OUT/rocprofT.log 2>&1
timeout 900 python tools/perf_survey.py > $OUT/perf_survey.log 2>&1; echo "exit $?" >> $OUT/perf_survey.log
cp gpurun_out/perf_survey.json $OUT/ 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_train -o tr -- python tools/runs/r3_train_prof.py > $OUT/rocprof_train.log 2>&1
timeout 200 python tools/archive/runs/r3_train_trace.py > $OUT/train_trace.log 2>&1
timeout 300 python tools/archive/runs/r3_train_time.py > $OUT/train_time.log 2>&1
timeout 300 python tools/archive/runs/r3_dense_coop_ab.py > $OUT/dense_coop_ab.log 2>&1
timeout 200 python tools/archive/runs/r3_quad_rotate_ab.py > $OUT/quad_rotate_ab.log 2>&1
timeout 200 python tools/archive/runs/r3_nam_fused_ab.py > $OUT/nam_fused_ab.log 2>&1
timeout 200 python tools/archive/runs/r3_small_zero_copy_ab.py > $OUT/small_zero_copy_ab.log 2>&1
timeout 300 python tools/archive/runs/r3_server_ab.py > $OUT/server_ab.log 2>&1
timeout 300 python tools/archive/runs/r3_server_stress.py > $OUT/server_stress.log 2>&1
timeout 300 python tools/archive/runs/r3_server_mixed.py > $OUT/server_mixed.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_srv1 -o s1 -- python tools/archive/runs/r3_server_trace.py 1 > $OUT/rocprof_server_on.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_srv0 -o s0 -- python tools/archive/runs/r3_server_trace.py 0 > $OUT/rocprof_server_off.log 2>&1
# keep the merged directory small: drop the raw per-dispatch CSVs of the PMC passes (summaries stay)
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
du -sh $OUT; ls $OUT
for f in bench_driver bench; do echo "== $f"; grep '^{' $OUT/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps')}, d.get('settled'))
r=d['roofline']; print('roof',r['kernel_ms'],r['frac'],r['frac_issued'], r.get('mfma_util_pmc'))
for k,v in d.get('configs',{}).items(): print(k, round(v['kernel_ms']*1e3,1),'us', round(v['frac'],3), round(v['frac_issued'],3))
for k,v in d.get('end_to_end',{}).items():
    if isinstance(v, dict) and 'wall_ms' in v: print(k, round(v['wall_ms'],3), 'ms', '%.3g'%v['value'], v.get('frac_of_kernel_rate'))
print(d.get('explorer_round'))
"; done
cat $OUT/pmc_bench.md; cat $OUT/pmc_targets.md
