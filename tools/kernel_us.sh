#!/bin/bash
# usage (GPU box): tools/kernel_us.sh "<bench flags>" VAR=val ...   -- one bench line per environment setting, kernel times only
R=${GRAFT_REPO_ROOT:-$PWD}
FLAGS=$1; shift
for e in "" "$@"; do
  env $e python3 $R/bench.py --steps 500 --warmup 50 --repeats 2 --no-cpu-baseline --no-acting $FLAGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('[$e]', d['ms_per_step'], d['roofline']['kernel_us_event_incl_boundary'])"
done
