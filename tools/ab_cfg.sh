#!/bin/bash
# usage: ab_cfg.sh old.so rounds -- bench args...: old vs in-tree library on another configuration, same box
OLD=$1; R=$2; shift 3
for i in $(seq 1 $R); do
  for w in old new; do
    if [ $w = old ]; then export RVIP_LIB=$OLD; else unset RVIP_LIB; fi
    timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-fit --no-aux > /tmp/ab_$w.json 2> /tmp/ab_$w.err || echo FAIL $w
    python -c "import json; d=json.loads(open('/tmp/ab_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
