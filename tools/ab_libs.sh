#!/bin/bash
# usage: ab_libs.sh rounds lib1.so lib2.so ... ("-" = the in-tree library): bench.py per library, alternating, same box
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset RVIP_LIB; else export RVIP_LIB=$lib; fi
    timeout -k 10 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --no-aux > /tmp/ab.json 2> /tmp/ab.err || echo FAIL $lib
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); k=d['kernels']; print('$lib', d['value'], d['ms_per_step'], 'pairs', k.get('rvip_conv3x3_wgrad_dgrad', {}).get('ms_per_step'), 'apply', k.get('rvip_bn_bwd_apply', {}).get('ms_per_step'), 'bn_apply', k.get('rvip_bn_apply', {}).get('ms_per_step'), 'finalize', k.get('rvip_bn_stats_finalize', {}).get('ms_per_step'))"
  done
done
