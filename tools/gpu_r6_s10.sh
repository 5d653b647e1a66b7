#!/bin/bash
# Round 6 session 10: slab pipe A/B (timing + same bits) and the slab bit-identity tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s10; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_slab_pipe_ab.py > $OUT/slab_pipe.log 2>&1; echo "exit $?" >> $OUT/slab_pipe.log
grep -v amdgpu.ids $OUT/slab_pipe.log | cut -c1-200
( time timeout 1200 python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "slab or hidden or mlp" > $OUT/pytest.log 2>&1 ) 2>> $OUT/pytest.log
grep -v "^\.*$" $OUT/pytest.log | tail -8
