#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3s5; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(quiet=True); print('build ok')" > $OUT/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x -k "pipelined or mlp or ge_ or dense or sweep" > $OUT/pytest_dense.log 2>&1
echo "pytest exit: $?" >> $OUT/pytest_dense.log
grep -v "^$" $OUT/pytest_dense.log | tail -25
timeout 400 python tools/archive/runs/r3_dense_pipe_ab.py > $OUT/dense_pipe_ab.log 2>&1
grep "dense_pipe" $OUT/dense_pipe_ab.log | python -c "
import sys, ast
for ln in sys.stdin:
    try: d = ast.literal_eval(ln.strip())
    except Exception: print(ln[:200]); continue
    print('%-52s %8.2f us  %.3f' % (d['what'], d['kernel_ms']*1e3, d['frac_mfma_peak']))
"
tail -3 $OUT/dense_pipe_ab.log | cut -c1-300
