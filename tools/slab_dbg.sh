#!/bin/bash
case=$1; world=$2; out=$3
mkdir -p $out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 WORLD_SIZE=$world
pids=""
for ((r=0; r<world; r++)); do
  RANK=$r LOCAL_RANK=0 timeout -k 10 200 python -u tools/slab_dbg.py $case $out > $out/rank$r.log 2>&1 &
  pids="$pids $!"
done
rc=0
for p in $pids; do wait $p || rc=1; done
cat $out/rank0.log | tail -8
exit $rc
