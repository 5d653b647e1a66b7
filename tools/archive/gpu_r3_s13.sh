#!/bin/bash
# PMC passes over the training kernels (instruction mix / waits), counters only (no --kernel-trace with --pmc)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s13; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp
T="python $R/tools/runs/r3_train_prof.py"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -f csv -d $R/$OUT/pmc_sq -o t -- $T > $R/$OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR -f csv -d $R/$OUT/pmc_inst -o t -- $T > $R/$OUT/pmc_inst.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_WAVES SQ_ACTIVE_INST_MISC -f csv -d $R/$OUT/pmc_act -o t -- $T > $R/$OUT/pmc_act.log 2>&1
cd $R
python tools/summarize_pmc.py $OUT/pmc_train.json $OUT/pmc_train.md sq=$OUT/pmc_sq inst=$OUT/pmc_inst act=$OUT/pmc_act > $OUT/pmc_summary.log 2>&1
cat $OUT/pmc_train.md; tail -3 $OUT/pmc_summary.log; tail -3 $OUT/pmc_inst.log
find $OUT -name "*counter_collection.csv" -size +2M -delete
