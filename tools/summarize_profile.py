#!/usr/bin/env python3
"""Copy the judged parts of a tools/profile.sh run from gpurun_out/ into profiles/ and derive
per-launch HBM traffic (profiles/traffic.json) the way MI355X_MICROARCH.md prescribes:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE reads half the bytes of wide coalesced
reads on gfx950; both counters are in KiB and come from separate --pmc passes."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: summarize_profile.py <tag> [configs index]; files and traffic.json keys carry "configs<i>" = BASELINE.json configs[i]
tag, cfg = sys.argv[1], "configs" + (sys.argv[2] if len(sys.argv) > 2 else "2")
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pat):
    return sorted(glob.glob(os.path.join(src, pat)))[0]


rows = list(csv.DictReader(open(one("stats/*/*kernel_stats.csv"))))
keep = [r for r in rows if "prism::" in r["Name"]]
with open(os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(keep)
    other = sum(float(r["TotalDurationNs"]) for r in rows if "prism::" not in r["Name"])
    f.write(f'"(all non-prism kernels: torch fills/copies during setup)",,{other:.0f},,,,,\n')


def kname(full):
    """'void prism::iqn_post_kernel<false>(prism::IqnArgs, ...)' -> 'iqn_post_kernel'"""
    n = full.split("(")[0].replace("prism::", "").replace("void ", "").strip()
    return n.split("<")[0]


def counter(pat):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(pat))):
        if "prism::" in r["Kernel_Name"]:
            agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for k, v in agg.items()}


fetch, write = counter("fetch/*/*counter_collection.csv"), counter("write/*/*counter_collection.csv")
tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic[cfg] = {k: int((2 * fetch[k] + write.get(k, 0.0)) * 1024) for k in fetch if "init" not in k and "rebuild" not in k}
traffic[cfg + "_raw_KiB"] = {k: {"FETCH_SIZE": round(fetch[k], 1), "WRITE_SIZE": round(write.get(k, 0.0), 1)} for k in traffic[cfg]}
traffic.setdefault("_meta", {})[cfg] = tag          # which profile run the figures of this configuration come from
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
line = [ln for ln in open(os.path.join(src, "stats.log")) if ln.startswith("{")]
if line:
    open(os.path.join(dst, f"{tag}_{cfg}_bench_under_rocprof.json"), "w").write(line[-1])
for r in keep:
    print(f'{r["Name"].split("(")[0]:32s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Percentage"]:>6s} %')
print(json.dumps(traffic[cfg], indent=1))
