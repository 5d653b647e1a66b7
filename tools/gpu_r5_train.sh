#!/bin/bash
# Round 5: training sessions.  usage: gpu_r5_train.sh <tag> "<train_swizzle values to profile>" [skip_tests]
# the training tests, the train_swizzle A/B with phase timelines, rocprofv3 kernel stats + PMC of the GFP-length fit per form.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
TAG=${1:-r5train}; FORMS=${2:-3}
OUT=gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
if [ -z "$3" ]; then
  timeout 900 python -m pytest tests/test_train_native.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_train.log 2>&1
  echo "pytest exit: $?" >> $OUT/pytest_train.log
  tail -15 $OUT/pytest_train.log
fi
timeout 300 python tools/runs/r5_train_swizzle_ab.py > $OUT/train_swizzle_ab.log 2>&1
tail -16 $OUT/train_swizzle_ab.log
cd /tmp
SPEC=""
for swz in $FORMS; do
  timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof_swz$swz -o tr -- python $R/tools/runs/r5_train_gfp_prof.py $swz 2 > $R/$OUT/rocprof_swz$swz.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -f csv -d $R/$OUT/pmc_inst_swz$swz -o t -- python $R/tools/runs/r5_train_gfp_prof.py $swz 1 > $R/$OUT/pmc_inst_swz$swz.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -f csv -d $R/$OUT/pmc_sq_swz$swz -o t -- python $R/tools/runs/r5_train_gfp_prof.py $swz 1 > $R/$OUT/pmc_sq_swz$swz.log 2>&1
  SPEC="$SPEC inst$swz=$OUT/pmc_inst_swz$swz sq$swz=$OUT/pmc_sq_swz$swz"
done
cd $R
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
for swz in $FORMS; do cat $(find $OUT/prof_swz$swz -name "*kernel_stats.csv" | head -1) | head -6; done
python tools/summarize_pmc.py $OUT/pmc_train.json $OUT/pmc_train.md $SPEC > $OUT/summarize.log 2>&1
cat $OUT/pmc_train.md | cut -c1-400 | head -30
