#!/bin/bash
# SQ counters of the attention kernels on tools/prof_sca.py (the r01 recipe): two --pmc passes, summed per kernel.
# Run on the GPU box from the repo root:  bash tools/measure_pmc.sh > gpurun_out/r02_pmc_attention.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/bevr_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for W in "" "1"; do
  echo "# WIDE=${W:-0} (${W:+kaiming-initialised offset heads: keys over the full learned range}${W:-module default init})"
  export WIDE=$W
  [ -z "$W" ] && unset WIDE
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1$W -o p -- python3 $ROOT/tools/prof_sca.py > $OUT/p1$W.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2$W -o p -- python3 $ROOT/tools/prof_sca.py > $OUT/p2$W.log 2>&1
  grep TIMES $OUT/p1$W.log
  python3 $ROOT/tools/pmc_sum.py $OUT/p1$W
  python3 $ROOT/tools/pmc_sum.py $OUT/p2$W
done
