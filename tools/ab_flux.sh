#!/bin/bash
# parity tests, then A/B of the marching flux kernels / look-ahead RHS with kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
for v in 00 10 11; do
  export CFD_MARCH_FLUX=${v:0:1} CFD_RHS_LOOKAHEAD=${v:1:1}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_fx$v -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/ab_fx$v.json 2> gpurun_out/ab_fx$v.err || exit 1
done
