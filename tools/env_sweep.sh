#!/bin/bash
# usage: env_sweep.sh VAR v1 v2 ... : bench.py (resident inputs, graph) per value of an experiment switch, same box
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --no-aux > /tmp/sw.json 2> /tmp/sw.err || echo FAIL $v
  python -c "import json; d=json.loads(open('/tmp/sw.json').read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
