#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench.py flags]   -- one rocprofv3 --pmc pass over a short eager run
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-acting --no-graph --profile-steps 0 --repeats 1 ${@:3} > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$OUT/*/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "prism::" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("prism::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v[len(v)//3:]) / len(v[len(v)//3:])) for c, v in d.items()})
PY
