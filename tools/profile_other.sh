#!/bin/bash
# kernel-trace statistics of the cfg-4 / cfg-5 steps (bench.py's other_configs legs as headline runs): gpurun -- 'bash tools/profile_other.sh'
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-fit --no-aux --no-other-configs --no-roofline-pass --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats -d $OUT/c4 -o c4 --output-format csv -- python3 $ROOT/bench.py --dim 512 --filters 64 --depth 5 --precision fp16 --batch 8 $COMMON > $OUT/cfg4.json 2> $OUT/cfg4.err
rocprofv3 --kernel-trace --stats -d $OUT/c5 -o c5 --output-format csv -- python3 $ROOT/bench.py --dim 256 --filters 32 --depth 4 --frames 16 --precision bf16 --batch 4 $COMMON > $OUT/cfg5.json 2> $OUT/cfg5.err
cp $(find $OUT/c4 -name 'c4_kernel_stats.csv' | head -1) $OUT/cfg4_kernel_stats.csv
cp $(find $OUT/c5 -name 'c5_kernel_stats.csv' | head -1) $OUT/cfg5_kernel_stats.csv
rm -rf $OUT/c4 $OUT/c5
