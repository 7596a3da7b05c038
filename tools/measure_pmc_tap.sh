#!/bin/bash
# SQ counters of the tap kernels on tools/tap_check.py time: two --pmc passes.  Run on the GPU box from the repo root.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/bevr_pmc_tap; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/p1 -o p -- python3 $ROOT/tools/tap_check.py time > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2 -o p -- python3 $ROOT/tools/tap_check.py time > $OUT/p2.log 2>&1
grep TAP $OUT/p1.log
python3 $ROOT/tools/pmc_sum.py $OUT/p1
python3 $ROOT/tools/pmc_sum.py $OUT/p2
