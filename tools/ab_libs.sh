#!/bin/bash
# usage (GPU box): tools/ab_libs.sh "<bench flags>" <rounds> lib1.so lib2.so ...   ("tree" = the in-tree library)
# Alternates bench.py between builds of the library inside ONE gpurun call (boxes differ by +-1 us per step).
R=${GRAFT_REPO_ROOT:-$PWD}
FLAGS=$1; N=$2; shift 2
for i in $(seq $N); do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset PRISM_HIP_LIB; else export PRISM_HIP_LIB=$(realpath $lib); fi
    python3 $R/bench.py --steps 2000 --warmup 100 --repeats 5 --no-cpu-baseline --no-acting $FLAGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], d['roofline']['kernel_us_event_incl_boundary'])"
  done
done
