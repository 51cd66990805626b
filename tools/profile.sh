#!/bin/bash
# usage: tools/profile.sh <tag> [bench.py flags, e.g. --config 3]
# rocprofv3 evidence for profiles/: kernel stats of the bench command, then HBM traffic counters
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Run on the GPU box via gpurun.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 2000 --warmup 200 --repeats 5 --no-cpu-baseline --no-acting ${@:2} > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-acting --no-graph --profile-steps 0 ${@:2} > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-acting --no-graph --profile-steps 0 ${@:2} > $OUT/write.log 2>&1
# (the per-launch traces are tens of MB: gpurun only carries 64 MiB back)
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
tail -1 $OUT/stats.log | cut -c1-300
find $OUT -name "*.csv" | head -20
