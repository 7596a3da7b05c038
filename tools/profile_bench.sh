#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command; summary -> gpurun_out/<tag>_kernel_stats.csv (copy to profiles/)
# Run on the GPU box from the repo root:  bash tools/profile_bench.sh r03_a [extra bench args]
TAG=${1:-r05}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/bevr_prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --f32-steps 0 "$@" > $OUT/bench.log 2>&1
grep '"metric"' $OUT/bench.log > $ROOT/gpurun_out/${TAG}_bench.json
find $OUT -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/${TAG}_kernel_stats.csv \;
head -45 $ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
