#!/bin/bash
# time kernel variants (PRISM_DBG bitmask) with the bench's HIP-event profile
for m in 0 1 2 3 4 7; do
  echo -n "PRISM_DBG=$m: "
  PRISM_DBG=$m timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-graph --profile-every 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: v for k, v in d['roofline']['kernel_us'].items() if 'fwd' in k or 'bwd' in k})"
done
