#!/bin/bash
# Build and run one probe under tools/probes on the GPU box: tools/archive/gpu_probe.sh mailbox_probe2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probes/$1.hip -o /tmp/$1 2>&1 | tail -5
timeout 120 /tmp/$1 $2 $3 > gpurun_out/$1.log 2>&1
echo "exit: $?" >> gpurun_out/$1.log
cat gpurun_out/$1.log | tail -40
