#!/bin/bash
# A/B timing of tap-kernel variant builds (bevrender_amd/lib_var_*/): interleaved rounds in one GPU call.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
MODE=${MODE:-time}
for round in 1 2; do
  echo default; python3 $ROOT/tools/tap_check.py $MODE 2>&1 | grep TAP
  for d in $ROOT/bevrender_amd/lib_var_*; do
    echo $d; BEVRENDER_LIB=$d/libbevrender_hip.so python3 $ROOT/tools/tap_check.py $MODE 2>&1 | grep TAP
  done
done
