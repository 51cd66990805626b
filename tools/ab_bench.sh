#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <other libprism_hip.so> [rounds] [bench.py flags]
# Alternates bench.py between the in-tree library and another build of it (PRISM_HIP_LIB) inside ONE gpurun call:
# boxes differ by +-1 us per step, so only numbers taken on the same box, interleaved, compare.
R=${GRAFT_REPO_ROOT:-$PWD}
OTHER=$(realpath "$1"); N=${2:-3}
for i in $(seq $N); do
  for v in other tree; do
    if [ $v = other ]; then export PRISM_HIP_LIB=$OTHER; else unset PRISM_HIP_LIB; fi
    python3 $R/bench.py --steps 2000 --warmup 100 --repeats 5 --no-cpu-baseline --no-acting ${@:3} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['kernel_us'])"
  done
done
