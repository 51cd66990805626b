#!/bin/bash
# one-line summary of a bench.py run (GPU box): steps/s, ms/step, per-kernel us
python bench.py --no-cpu-baseline --steps ${STEPS:-2000} --warmup 200 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'])"
