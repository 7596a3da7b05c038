#!/bin/bash
# HBM traffic per launch of the path's kernels on the bench command: two rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), summed per kernel by tools/traffic_sum.py into gpurun_out/r05_traffic.json (the GPU
# box's only writable path that travels back; copy it to profiles/r05_traffic.json), stamped with the kernel sources'
# sha so that bench.py only attaches it to the kernels it was measured on.
# Run on the GPU box from the repo root:   bash tools/measure_traffic.sh [batch]
set -e
B=${1:-8}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/bevr_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --f32-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
SHA=$(cd $ROOT && python3 -c "import bench; print(bench.csrc_sha())")
python3 $ROOT/tools/traffic_sum.py $OUT $B "$CMD" $SHA > $ROOT/gpurun_out/r05_traffic.json
cat $ROOT/gpurun_out/r05_traffic.json
