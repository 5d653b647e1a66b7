#!/bin/bash
# Kernel timeline of the one-rank distributed bench (RCCL all-gather overlapped with compute) + L=14 A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True)" > $OUT/env.log 2>&1
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
rm -rf $OUT/trace_dist
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace_dist -o t -- python bench.py --gpus 1 --steps 30 --warmup 5 --force-dist --no-cpu-baseline > $OUT/trace_dist.log 2>&1
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -k "mlp_ge or hidden or config3" > $OUT/pytest_l14.log 2>&1
timeout 300 python - > $OUT/l14_ab.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.'); sys.argv=['x','none']
import tools.perf_survey as ps
for v in ():
    ps.time_score("cnn", 14, "UGCA", 100, 3, 100_000, 32, 5, variant=v, label=f"cnn L=14 M=3 N=1e5 variant {v}")
for v in ():
    ps.time_score("cnn", 14, "UGCA", 100, 3, 1_000_000, 32, 5, variant=v, reps=3, label=f"cnn L=14 M=3 N=1e6 variant {v}")
PY
tail -2 $OUT/pytest_l14.log; tail -6 $OUT/l14_ab.log | cut -c1-200; ls $OUT/trace_dist
timeout 300 python - > $OUT/ge_mlp_ab.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import tools.perf_survey as ps
AAS = ps.AAS
ps.time_score("mlp", 14, "UGCA", 100, 1, 100_000, label="C3 mlp L=14 H=100 M=1 N=1e5")
ps.time_score("mlp", 14, "UGCA", 100, 1, 1_000_000, reps=5, label="mlp L=14 N=1e6")
ps.time_score("mlp", 90, AAS, 100, 1, 100_000, reps=5, label="mlp L=90 A=20 N=1e5")
ps.time_score("ge", 90, AAS, 100, 8, 100_000, label="C4 ge L=90 A=20 M=8 N=1e5")
ps.time_score("ge", 90, AAS, 100, 8, 1_000_000, reps=5, label="ge M=8 N=1e6")
ps.time_score("ge", 90, AAS, 100, 1, 100_000, label="ge M=1 N=1e5")
PY
cat $OUT/ge_mlp_ab.log | cut -c1-160
grep '^{' $OUT/trace_dist.log | tail -1 | cut -c1-330
