#!/bin/bash
# GPU box: stream-phase stamps of the forward tile for experimental builds (tools/ubench/libexp_*.so, wrong numerics)
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in tree $R/tools/ubench/libexp_*.so; do
  if [ $lib = tree ]; then unset PRISM_HIP_LIB; else export PRISM_HIP_LIB=$lib; fi
  echo "== $(basename $lib)"
  python3 $R/tools/stamp_profile.py 2 2>&1 | grep -E "streamed phi|partials out|workgroup total" | head -3
done
