#!/bin/bash
# One probe under tools/probes: build on the box, run, log to gpurun_out/r4_<name>.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/$1.hip -o /tmp/$1 > /dev/null 2>&1
timeout 150 /tmp/$1 > gpurun_out/r4_$1.log 2>&1; echo "exit: $?" >> gpurun_out/r4_$1.log
cat gpurun_out/r4_$1.log
