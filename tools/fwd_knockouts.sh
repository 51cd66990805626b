#!/bin/bash
# usage (build container): tools/fwd_knockouts.sh build   -- variants of the library with parts of the forward tile knocked out
#        (GPU box):        tools/fwd_knockouts.sh run     -- rocprofv3 kernel averages of each variant, same box
# Knock-outs give WRONG numbers on purpose; they price a part of fwd_tile_kernel's streamed phase (FW_KO in fwd_kernels.h):
#   1 no ReLU(phi) save inside the stream, 2 no three-piece split of the trunk operand, 4 half the weight bytes at the same MFMAs
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p prism_amd/csrc/_exp
  for ko in 1 2 4 3 7; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Iprism_amd/csrc -Wno-unused-function \
      -DFW_KO=$ko -c prism_amd/csrc/learner.hip -o prism_amd/csrc/_exp/learner_ko$ko.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o prism_amd/csrc/_exp/lib_ko$ko.so prism_amd/csrc/_exp/learner_ko$ko.o \
      prism_amd/csrc/_build/api.o prism_amd/csrc/_build/profile.o prism_amd/csrc/_build/replay.o prism_amd/csrc/_build/direct.o &
  done
  wait
  ls -la prism_amd/csrc/_exp/*.so
else
  for rep in 1 2; do
    unset PRISM_HIP_LIB; tools/kstats.sh ko0_$rep
    for ko in 1 2 4 3 7; do PRISM_HIP_LIB=$R/prism_amd/csrc/_exp/lib_ko$ko.so tools/kstats.sh ko${ko}_$rep; done
  done
fi
