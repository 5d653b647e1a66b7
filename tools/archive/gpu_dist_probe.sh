#!/bin/bash
# A/B of the overlapped all-gather on one rank: CUs reserved for RCCL vs none.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
for r in 0 2 4 8; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$r bench.py --gpus 1 --steps 100 --warmup 10 --force-dist --no-cpu-baseline --reserve-cus $r > $OUT/bench_dist_r$r.log 2>&1
  echo "exit $?" >> $OUT/bench_dist_r$r.log
done
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_nodist.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp or dense" > $OUT/pytest_mlp.log 2>&1
for r in 0 2 4 8; do grep '^{' $OUT/bench_dist_r$r.log | tail -1 | cut -c1-140; done
