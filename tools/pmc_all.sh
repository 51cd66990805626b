#!/bin/bash
# usage (GPU box): tools/pmc_all.sh <tag> config <i>          -- the five rocprofv3 --pmc passes DESIGN.md quotes, bench.py --config i
#                  tools/pmc_all.sh <tag> preset "<case name>" -- the same over one ablation preset (tools/bench_presets.py)
# Separate passes, no trace domains beside --kernel-trace (MI355X_MICROARCH.md).  Then, in the build container:
#   python tools/summarize_pmc.py <tag> 5 <i | preset label>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KIND=$2; WHAT=$3
SETS=("SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
      "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum")
for k in 0 1 2 3 4; do
  OUT=$R/gpurun_out/pmc_${TAG}_$k
  rm -rf $OUT && mkdir -p $OUT
  if [ "$KIND" = config ]; then
    rocprofv3 --kernel-trace --pmc ${SETS[$k]} --output-format csv -d $OUT -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-acting --no-graph --profile-steps 0 --repeats 1 --config $WHAT > $OUT/run.log 2>&1
  else
    PRESET_STEPS=40 PRESET_EAGER=1 rocprofv3 --kernel-trace --pmc ${SETS[$k]} --output-format csv -d $OUT -- python3 $R/tools/bench_presets.py "$WHAT" > $OUT/run.log 2>&1
  fi
  find $OUT -name "*kernel_trace.csv" -delete
  find $OUT -name "*agent_info.csv" -delete
  echo "pass $k done: $(find $OUT -name '*counter_collection.csv' | wc -l) file(s)"
done
