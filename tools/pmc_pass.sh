#!/bin/bash
# usage (GPU box): bash tools/pmc_pass.sh TAG "COUNTER1 COUNTER2 ..." : per-kernel means of a counter set over eager steps -> gpurun_out/TAG.txt
set -e -o pipefail
TAG=$1; CNT=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT -d $OUT/p -o m --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-fit --no-aux --no-roofline-pass --no-graph --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/err.txt
M=$(find $OUT/p -name 'm_counter_collection.csv' | head -1); MT=$(find $OUT/p -name 'm_kernel_trace.csv' | head -1)
python3 $ROOT/tools/pmc_kernels.py "$M" "$MT" > $ROOT/gpurun_out/$TAG.txt
rm -rf $OUT/p
head -3 $ROOT/gpurun_out/$TAG.txt | cut -c1-200
