#!/bin/bash
# usage: ab_env.sh rounds VAR v1 v2 ...: bench.py per value of the environment variable VAR ("-" = unset), alternating, same box
R=$1; VAR=$2; shift; shift
for i in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --no-aux --no-other-configs > /tmp/ab.json 2> /tmp/ab.err || echo FAIL $v
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); k=d['kernels']; print('$VAR=$v', d['value'], d['ms_per_step'], ' '.join('%s %.4f' % (n.replace('rvip_',''), k[n]['ms_per_step']) for n in ('rvip_conv3x3_fwd','rvip_conv3x3_fwd_stats','rvip_conv3x3_wgrad_dgrad','rvip_bn_bwd_apply','rvip_bn_apply') if n in k))"
  done
done
