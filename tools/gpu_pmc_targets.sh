#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
rm -rf $OUT/pmcT_*
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmcT_sq -o t -- python tools/pmc_targets.py > $OUT/pmcT_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $OUT/pmcT_inst -o t -- python tools/pmc_targets.py > $OUT/pmcT_inst.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -f csv -d $OUT/pmcT_act -o t -- python tools/pmc_targets.py > $OUT/pmcT_act.log 2>&1
ls $OUT/pmcT_*/ ; tail -3 $OUT/pmcT_act.log | cut -c1-200
