#!/bin/bash
# usage (GPU box): bash tools/timeline_run.sh TAG [bench args]: kernel trace of the captured step -> gpurun_out/TAG_timeline.txt
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-fit --no-aux --no-roofline-pass --steps 30 --warmup 5 --dump-labels $OUT/labels.json "$@" > $OUT/bench.json 2> $OUT/err.txt
T=$(find $OUT/tr -name 't_kernel_trace.csv' | head -1)
python3 $ROOT/tools/step_timeline.py "$T" $OUT/labels.json > $ROOT/gpurun_out/${TAG}_timeline.txt
rm -rf $OUT/tr
tail -3 $ROOT/gpurun_out/${TAG}_timeline.txt
