#!/bin/bash
# Training step: padded strides, rows per slice A/B, timeline with staging, kernel stats per rows setting.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4train2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/archive/runs/r3_train_trace.py > $OUT/train_trace.log 2>&1
grep -v amdgpu $OUT/train_trace.log | head -22
timeout 300 python tools/archive/runs/r4_train_rows.py > $OUT/train_rows.log 2>&1
grep -v amdgpu $OUT/train_rows.log
