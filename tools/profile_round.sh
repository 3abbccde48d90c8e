#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the bench command,
#   2. one --pmc pass per counter (FETCH_SIZE, WRITE_SIZE) of a short run of the same command.
# Usage: bash tools/profile_round.sh <tag> [bench.py arguments, e.g. --mesh graded]
#        -> gpurun_out/<tag>_{stats,fetch,write}/, gpurun_out/<tag>_bench.json
# The program itself follows `--` (no env / bash -c hop under the profiler).
tag=${1:-rX}
shift
extra="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- python3 bench.py --steps 10 --warmup 3 --settle 0 --no-variants $extra > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-variants --settle 0 $extra > gpurun_out/${tag}_fetch.json 2> gpurun_out/${tag}_fetch.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-variants --settle 0 $extra > gpurun_out/${tag}_write.json 2> gpurun_out/${tag}_write.err || exit 1
echo done
