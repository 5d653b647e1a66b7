#!/bin/bash
# Round 6 session 15: slab + pair rows -- A/B and the dense tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s15; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_slab_pair_ab.py > $OUT/slab_pair.log 2>&1; echo "exit $?" >> $OUT/slab_pair.log
grep -v amdgpu.ids $OUT/slab_pair.log | cut -c1-250
( time timeout 1200 python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py tests/test_gpu_resident.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp or slab or hidden or dense or pair or small" > $OUT/pytest.log 2>&1 ) 2>> $OUT/pytest.log
grep -v "^\.*$" $OUT/pytest.log | tail -12
