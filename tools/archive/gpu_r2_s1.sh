#!/bin/bash
# Round 2, GPU session 1: parity after the distributed / bench rework, driver-style bench, probe 1.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "exit $?" >> $OUT/bench_driver.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; echo "exit $?" >> $OUT/bench.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --no-cpu-baseline > $OUT/bench_dist1.log 2>&1; echo "exit $?" >> $OUT/bench_dist1.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --mode member --no-cpu-baseline --no-extras > $OUT/bench_dist1_member.log 2>&1; echo "exit $?" >> $OUT/bench_dist1_member.log
timeout 400 python tools/archive/runs/r2_probe1.py > $OUT/probe1.log 2>&1; echo "exit $?" >> $OUT/probe1.log
tail -3 $OUT/pytest_gpu.log
for f in bench_driver bench bench_dist1 bench_dist1_member; do echo "== $f"; tail -c 1500 $OUT/$f.log; done
grep what $OUT/probe1.log | python3 -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l.strip()); print(d['what'], round(d['kernel_ms']*1e3,1),'us', round(d['frac_mfma_peak'],3))
"
