#!/bin/bash
# PMC passes (HBM bytes per kernel) and kernel stats of the k-omega variant of the bench workload (run through gpurun).
# Usage: bash tools/profile_komega.sh <tag>   -> gpurun_out/<tag>_{stats,fetch,write}/
tag=${1:-rXk}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- python3 bench.py --turbulent --steps 10 --warmup 3 --no-cpu --settle 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o f -- python3 bench.py --turbulent --steps 1 --warmup 1 --no-cpu --settle 0 > gpurun_out/${tag}_fetch.json 2> gpurun_out/${tag}_fetch.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o w -- python3 bench.py --turbulent --steps 1 --warmup 1 --no-cpu --settle 0 > gpurun_out/${tag}_write.json 2> gpurun_out/${tag}_write.err || exit 1
echo done
