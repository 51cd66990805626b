#!/usr/bin/env python3
"""Collect the per-kernel counter averages of several tools/pmc.sh passes (gpurun_out/pmc_<tag>_<k>/) into
profiles/<tag>_configs<i>_pmc.json, with the derived per-CU / per-SIMD figures DESIGN.md quotes.
usage: summarize_pmc.py <tag> <n passes> [configs index]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n = sys.argv[1], int(sys.argv[2])
_c = sys.argv[3] if len(sys.argv) > 3 else "2"
cfg = "configs" + _c if _c.isdigit() else _c          # BASELINE.json configs[i], or a label such as preset_additive
kern = collections.defaultdict(dict)
for k in range(n):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{k}", "*", "*counter_collection.csv")))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "prism::" in r["Kernel_Name"] and "init" not in r["Kernel_Name"] and "rebuild" not in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("prism::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, d in agg.items():
        for c, v in d.items():
            kern[name][c] = round(sum(v[len(v) // 3:]) / len(v[len(v) // 3:]))
CUS, SIMDS = 256, 1024
for name, d in kern.items():
    der = {}
    if "SQ_BUSY_CU_CYCLES" in d:
        der["cycles_per_CU"] = round(d["SQ_BUSY_CU_CYCLES"] / CUS)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
        der["mfma_busy_per_SIMD"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS)
        der["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / (d["SQ_BUSY_CU_CYCLES"] / CUS), 3)
    if "SQ_ACTIVE_INST_VALU" in d:
        der["valu_cycles_per_SIMD"] = round(4 * d["SQ_ACTIVE_INST_VALU"] / SIMDS)
    if "TA_TA_BUSY_sum" in d and d.get("SQ_BUSY_CU_CYCLES"):
        der["ta_busy_frac"] = round(d["TA_TA_BUSY_sum"] / CUS / (d["SQ_BUSY_CU_CYCLES"] / CUS), 3)
    if "TCP_TCC_READ_REQ_sum" in d:
        der["l2_read_MB"] = round(d["TCP_TCC_READ_REQ_sum"] * 128 / 1e6, 1)
        if d.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            der["l1_hit_frac"] = round(1 - d["TCP_TCC_READ_REQ_sum"] / d["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
    if "SQ_LDS_IDX_ACTIVE" in d and d.get("SQ_BUSY_CU_CYCLES"):
        der["lds_active_frac"] = round(d["SQ_LDS_IDX_ACTIVE"] / CUS / (d["SQ_BUSY_CU_CYCLES"] / CUS), 3)
        if "SQ_LDS_BANK_CONFLICT" in d:
            der["lds_conflict_frac_of_active"] = round(d["SQ_LDS_BANK_CONFLICT"] / max(1, d["SQ_LDS_IDX_ACTIVE"]), 3)
    d["derived"] = der
out = {"_what": f"rocprofv3 --pmc, {n} separate passes (tools/pmc.sh), bench.py --steps 40 --no-graph, {cfg}; averages per launch, "
                "summed over the chip (256 CUs, 1024 SIMDs). SQ_ACTIVE_INST_* and SQ_WAIT_* count quad-cycles.",
       "kernels": kern}
dst = os.path.join(ROOT, "profiles", f"{tag}_{cfg}_pmc.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for name, d in kern.items():
    print(name, d["derived"])
