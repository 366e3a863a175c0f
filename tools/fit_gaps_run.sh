#!/bin/bash
# usage (GPU box): bash tools/fit_gaps_run.sh TAG -> gpurun_out/TAG_fit_gaps.txt
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tr -o t --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-aux --no-roofline-pass --steps 60 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/err.txt
T=$(find $OUT/tr -name 't_kernel_trace.csv' | head -1); M=$(find $OUT/tr -name 't_memory_copy_trace.csv' | head -1)
python3 $ROOT/tools/fit_gaps.py "$T" $M > $ROOT/gpurun_out/${TAG}_fit_gaps.txt
head -3 "$M" > $ROOT/gpurun_out/${TAG}_memcopy_head.txt || true
rm -rf $OUT/tr
cat $ROOT/gpurun_out/${TAG}_fit_gaps.txt
