#!/bin/bash
# usage (GPU box): tools/kstats_preset.sh <tag> "<substring of a tools/bench_presets.py case name>"
# rocprofv3 kernel durations (average, us) of one ablation preset's steps; PRISM_HIP_LIB / PRISM_GEMM as for kstats.sh.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ksp_$1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_presets.py "$2" > $OUT/run.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
grep "steps/s" $OUT/run.log
python3 - $OUT $1 <<'P'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[0]
out = []
for r in csv.DictReader(open(f)):
    if "prism::" in r["Name"] and int(r["Calls"]) > 100:
        out.append((r["Name"].split("(")[0].replace("void ", "").replace("prism::", "")[:40], round(float(r["AverageNs"]) / 1e3, 2)))
print(sys.argv[2], " ".join(f"{k}={v}" for k, v in out), "sum", round(sum(v for _, v in out), 2))
P
