#!/bin/bash
# Round 4, session 7: three tiles per resident workgroup (serve_quads): resident tests + A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s7; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "resident or small_call or mailbox" > $OUT/pytest_resident.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_resident.log
grep -v "^\.*$" $OUT/pytest_resident.log | tail -30
timeout 500 python tools/archive/runs/r4_server_quads_ab.py > $OUT/quads_ab.log 2>&1; echo "exit: $?" >> $OUT/quads_ab.log
grep -v "amdgpu.ids" $OUT/quads_ab.log
