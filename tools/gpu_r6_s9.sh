#!/bin/bash
# Round 6 session 9: phase timeline of the slab form (make trace-phases build).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s9; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
( time make -C flexs_amd/csrc trace-phases -j16 > $OUT/make.log 2>&1 ) 2>> $OUT/make.log; tail -3 $OUT/make.log
FX_SET=r6p FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace_phases.so timeout 600 python tools/trace_probe.py > $OUT/trace.log 2>&1
cp gpurun_out/trace_probe_r6p.json $OUT/ 2>/dev/null; tail -2 $OUT/trace.log | cut -c1-300
