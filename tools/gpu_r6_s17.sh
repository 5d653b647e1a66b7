#!/bin/bash
# Round 6 session 17: position-major first layer of the protein MLP -- A/B + tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s17; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_protein_mlp_wide.py > $OUT/ab.log 2>&1; echo "exit $?" >> $OUT/ab.log
grep -v amdgpu.ids $OUT/ab.log | cut -c1-250
( time timeout 1200 python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp or dense or hidden or position" > $OUT/pytest.log 2>&1 ) 2>> $OUT/pytest.log
grep -v "^\.*$" $OUT/pytest.log | tail -15
