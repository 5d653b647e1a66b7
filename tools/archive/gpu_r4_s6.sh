#!/bin/bash
# Round 4, session 6: layer-parallel protein small-batch form: parity tests, A/B; plus the tests that failed in session 5.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s6; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "layer_parallel or pair_segmented or l14_unrolled or mlp_small_launch_form or population or decode_score or protein" > $OUT/pytest_lp.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_lp.log
grep -v "^\.*$" $OUT/pytest_lp.log | tail -30
timeout 300 python tools/archive/runs/r4_protein_small.py > $OUT/protein_small.log 2>&1; echo "exit: $?" >> $OUT/protein_small.log
grep -v "amdgpu.ids" $OUT/protein_small.log
