#!/bin/bash
# GPU session 2: re-check parity after kernel changes, bench variants, full bench line,
# kernel trace (csv), PMC passes (separate runs, no tracing domains), perf survey.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1

timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log

for v in 1 4 5; do
  timeout 200 python bench.py --steps 100 --warmup 10 --variant $v --no-cpu-baseline > $OUT/bench_v$v.log 2>&1
  echo "exit $?" >> $OUT/bench_v$v.log
done
timeout 400 python bench.py > $OUT/bench.log 2>&1
echo "exit $?" >> $OUT/bench.log

rm -rf $OUT/prof $OUT/pmc*
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o r1 -- python bench.py --no-cpu-baseline > $OUT/rocprof.log 2>&1
# PMC: one counter group per run, --pmc only (no trace domains)


timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1

find $OUT/prof $OUT/pmc_* -type f | head -40 > $OUT/files.log

timeout 900 python tools/perf_survey.py > $OUT/perf_survey.log 2>&1
echo "exit $?" >> $OUT/perf_survey.log
tail -2 $OUT/pytest_gpu.log; tail -2 $OUT/bench.log | cut -c1-400
# the N>1 code path (RCCL init + all-gather + max-over-ranks) on one rank
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --reserve-cus 4 --no-cpu-baseline > $OUT/bench_dist1.log 2>&1
echo "exit $?" >> $OUT/bench_dist1.log
