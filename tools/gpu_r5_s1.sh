#!/bin/bash
# Round 5, session 1: the training tests (GFP-length step against the oracle under every train_swizzle form, the un-gated
# bit-identity test), the A/B timing with phase timelines, rocprofv3 kernel stats + PMC of the GFP-length fit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=gpurun_out/r5s1; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests/test_train_native.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_train.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_train.log
tail -15 $OUT/pytest_train.log
timeout 300 python tools/runs/r5_train_swizzle_ab.py > $OUT/train_swizzle_ab.log 2>&1
tail -12 $OUT/train_swizzle_ab.log
cd /tmp
for swz in 0 2; do
  timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof_swz$swz -o tr -- python $R/tools/runs/r5_train_gfp_prof.py $swz 2 > $R/$OUT/rocprof_swz$swz.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -f csv -d $R/$OUT/pmc_inst_swz$swz -o t -- python $R/tools/runs/r5_train_gfp_prof.py $swz 1 > $R/$OUT/pmc_inst_swz$swz.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $R/$OUT/pmc_sq_swz$swz -o t -- python $R/tools/runs/r5_train_gfp_prof.py $swz 1 > $R/$OUT/pmc_sq_swz$swz.log 2>&1
done
cd $R
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
for swz in 0 2; do cat $(find $OUT/prof_swz$swz -name "*kernel_stats.csv" | head -1) | head -6; done
python tools/summarize_pmc.py $OUT/pmc_train.json $OUT/pmc_train.md inst0=$OUT/pmc_inst_swz0 sq0=$OUT/pmc_sq_swz0 inst2=$OUT/pmc_inst_swz2 sq2=$OUT/pmc_sq_swz2 > $OUT/summarize.log 2>&1
cat $OUT/pmc_train.md | cut -c1-400 | head -30
