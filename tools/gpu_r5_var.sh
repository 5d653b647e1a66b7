#!/bin/bash
# fit time + phase timeline of the GFP-length ensemble for each experiment build (flexs_amd/libflexs_amd_v*.so) and the default library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for lib in "" $(ls flexs_amd/libflexs_amd_v*.so 2>/dev/null); do
  if [ -z "$lib" ]; then python tools/runs/r5_train_fit_time.py $1; else FLEXS_AMD_LIB=$PWD/$lib python tools/runs/r5_train_fit_time.py $1; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_var.log
