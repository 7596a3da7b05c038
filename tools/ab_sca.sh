#!/bin/bash
# A/B timing of variant builds (bevrender_amd/lib_var_*/) on the SCA block (tools/prof_sca.py): interleaved rounds in one GPU call.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do
  echo default; timeout -k 10 200 python3 $ROOT/tools/prof_sca.py 2>&1 | grep TIMES
  for d in $ROOT/bevrender_amd/lib_var_*; do
    echo $d; BEVRENDER_LIB=$d/libbevrender_hip.so timeout -k 10 200 python3 $ROOT/tools/prof_sca.py 2>&1 | grep TIMES
  done
done
