#!/bin/bash
# A/B timing of cell-kernel variant builds (bevrender_amd/lib_var_*/): interleaved rounds in one GPU call.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do
  python3 $ROOT/tools/prof_cell.py 2>&1 | grep TIMES
  for d in $ROOT/bevrender_amd/lib_var_*; do
    BEVRENDER_LIB=$d/libbevrender_hip.so python3 $ROOT/tools/prof_cell.py 2>&1 | grep TIMES
  done
done
