#!/bin/bash
# Round 6 session 11: slab variants (separate library builds made in the container: libflexs_amd_v_*.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6s11; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for v in base dma2 dma3 dma4 reg3; do
  FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_v_$v.so timeout 300 python tools/runs/r6_slab_variants.py 2>&1 | grep -v amdgpu.ids >> $OUT/variants.log
done; done
sort -k2,8 -s $OUT/variants.log | cut -c1-120
