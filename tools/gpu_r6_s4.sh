#!/bin/bash
# Round 6 session 4: slab-form leftover tiles (A/B + bit identity), hostile-host tests after the relay-sequence fix.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s4; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python tools/runs/r6_slab_coop_ab.py > $OUT/slab_coop_ab.log 2>&1; echo "exit $?" >> $OUT/slab_coop_ab.log
grep -v amdgpu.ids $OUT/slab_coop_ab.log | cut -c1-260
( time timeout 900 python -m pytest tests/test_gpu_forms.py tests/test_gpu_hostile_host.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "slab or hostile or hidden or stopped or idle or pinned or foreign" > $OUT/pytest_s4.log 2>&1 ) 2>> $OUT/pytest_s4.log
grep -v "^\.*$" $OUT/pytest_s4.log | tail -40
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_hostile_host.py -m gpu -q --tb=line -p no:cacheprovider --timeout 600 2>&1 | tail -3; done
