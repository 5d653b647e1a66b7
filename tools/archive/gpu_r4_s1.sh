#!/bin/bash
# Round 4, session 1: probes for the wide resident form and the MFMA shape question, then the GPU suite and the driver's bench command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4s1; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
for p in mailbox_probe3 mfma_shape_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/$p.hip -o /tmp/$p > /dev/null 2>&1
  timeout 120 /tmp/$p > $OUT/$p.log 2>&1; echo "exit: $?" >> $OUT/$p.log
done
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "exit $?" >> $OUT/bench_driver.log
tail -5 $OUT/pytest_gpu.log
cat $OUT/mailbox_probe3.log
cat $OUT/mfma_shape_probe.log
grep '^{' $OUT/bench_driver.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps')})
print(json.dumps(d['roofline'].get('per_config'),indent=0))
print(json.dumps(d['config'].get('path'),indent=0))
"
