#!/bin/bash
# usage (GPU box): tools/kstats.sh <tag> [bench.py flags]  -- rocprofv3 kernel durations (average, us) of a short bench run.
# PRISM_HIP_LIB=<path> in the environment selects another build of the library (same-box A/B).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ks_$1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 1000 --warmup 100 --repeats 2 --no-cpu-baseline --no-acting ${@:2} > $OUT/run.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python3 - $OUT $1 <<'P'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[0]
out = []
for r in csv.DictReader(open(f)):
    if "prism::" in r["Name"] and int(r["Calls"]) > 100:
        out.append((r["Name"].split("(")[0].replace("void ", "").replace("prism::", "")[:40], round(float(r["AverageNs"]) / 1e3, 2)))
print(sys.argv[2], " ".join(f"{k}={v}" for k, v in out), "sum", round(sum(v for _, v in out), 2))
P
