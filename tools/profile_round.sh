#!/bin/bash
# Round profile on the GPU box (one MI355X): kernel-trace statistics of the benchmark command, the two HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE in runs of their own, --kernel-trace only), an MFMA-busy pass, and the summaries bench.py / DESIGN.md cite.
#   gpurun -- 'bash tools/profile_round.sh r05'
set -e -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-fit --no-aux --no-other-configs --no-roofline-pass"      # only training steps in the traces
# 1. the benchmarked command itself (captured step), per-kernel durations
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- $BENCH --steps 50 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# 2./3. HBM traffic: eager launches (one dispatch per kernel), few steps; the step count comes from the bench line itself
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f --output-format csv -- $BENCH --no-graph --steps 3 --warmup 1 > $OUT/pmc_f.json 2> $OUT/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w --output-format csv -- $BENCH --no-graph --steps 3 --warmup 1 > $OUT/pmc_w.json 2> $OUT/pmc_w.err
# 4. matrix-pipe occupancy of every kernel
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_m -o m --output-format csv -- $BENCH --no-graph --steps 3 --warmup 1 > $OUT/pmc_m.json 2> $OUT/pmc_m.err || echo "mfma pass failed (counter set)" >&2
F=$(find $OUT/pmc_f -name 'f_counter_collection.csv' | head -1); W=$(find $OUT/pmc_w -name 'w_counter_collection.csv' | head -1); T=$(find $OUT/pmc_f -name 'f_kernel_trace.csv' | head -1)
STEPS=$(python3 -c "
import json, sys
line = [l for l in open('$OUT/pmc_f.json') if l.lstrip().startswith('{')][-1]
print(json.loads(line)['steps_executed_total'])")
python3 $ROOT/tools/pmc_summary.py --steps $STEPS "$F" "$W" "$T" > $OUT/${TAG}_pmc_summary.json
M=$(find $OUT/pmc_m -name 'm_counter_collection.csv' | head -1); MT=$(find $OUT/pmc_m -name 'm_kernel_trace.csv' | head -1)
[ -n "$M" ] && python3 $ROOT/tools/pmc_kernels.py "$M" "$MT" > $OUT/${TAG}_pmc_mfma.txt || true
S=$(find $OUT/stats -name 'st_kernel_stats.csv' | head -1)
cp "$S" $OUT/${TAG}_bench_kernel_stats.csv
python3 - <<PY > $OUT/${TAG}_bench_kernel_stats.csv.meta.json
import json, sys
sys.path.insert(0, "$ROOT/tools")
import pmc_summary
print(json.dumps({"csrc_sha16": pmc_summary.csrc_sha16(), "workload": "dim256_f32_d4_b32_t0_bf16", "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-fit --no-aux --no-other-configs --no-roofline-pass --steps 50 --warmup 5"}))
PY
# keep the merge-back small: the raw traces stay on the box
rm -rf $OUT/stats $OUT/pmc_f/*/*kernel_trace* 2>/dev/null || true
du -sh $OUT
python3 - <<PY
import json
d = json.load(open('$OUT/${TAG}_pmc_summary.json'))
print('hbm_bytes_per_step %.2f GB over %d kernels' % (d['hbm_bytes_per_step'] / 1e9, len(d['kernels'])))
PY
