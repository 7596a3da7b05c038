#!/bin/bash
# SQ counters of the query-side backward over scattered keys -- the slab kernel and the query-tile kernel it replaces -- on
# tools/prof_sca.py (B = 1, 6 views): --pmc passes only (no trace domains).  Run on the GPU box from the repo root;
# output -> gpurun_out/r05_pmc_slab.txt (copy to profiles/).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/bevr_pmc_slab; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export ITERS=1
for g in 1 0; do
export BEVR_SLAB=$g
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $OUT/p1_$g -o p -- python3 $ROOT/tools/prof_sca.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/p2_$g -o p -- python3 $ROOT/tools/prof_sca.py > $OUT/p2.log 2>&1
echo "BEVR_SLAB=$g"
python3 $ROOT/tools/pmc_sum.py $OUT/p1_$g | grep "attn_slab_bwd_q\|attn_bwd_q_kernel"
python3 $ROOT/tools/pmc_sum.py $OUT/p2_$g | grep "attn_slab_bwd_q\|attn_bwd_q_kernel"
done
