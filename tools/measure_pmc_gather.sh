#!/bin/bash
# SQ counters of the scattered-key forward kernels (gather and region) on tools/prof_sca.py: --pmc passes.  Run on the GPU
# box from the repo root.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/bevr_pmc_gather; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export ITERS=1 FWD_ONLY=1
for g in 1 0; do
export BEVR_GATHER=$g
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/p1_$g -o p -- python3 $ROOT/tools/prof_sca.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2_$g -o p -- python3 $ROOT/tools/prof_sca.py > $OUT/p2.log 2>&1
echo "BEVR_GATHER=$g"
python3 $ROOT/tools/pmc_sum.py $OUT/p1_$g | grep "attn_gather_fwd\|attn_fwd_kernel"
python3 $ROOT/tools/pmc_sum.py $OUT/p2_$g | grep "attn_gather_fwd\|attn_fwd_kernel"
done
