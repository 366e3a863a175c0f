"""Summarise rocprofv3 PMC passes into profiles/<round>_pmc_summary.json: HBM bytes per launch and per step for EVERY kernel of
the step (grouped by device-function name with its template arguments stripped), and the digest of the kernel sources they
were taken on (bench.py refuses a summary whose digest differs from the tree's).

Usage (on the GPU box; each counter in its own run, --kernel-trace only; MI355X_MICROARCH.md section HBM):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps S --warmup 2 --no-cpu-baseline --no-graph --no-fit --no-aux
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py (same)
    python3 tools/pmc_summary.py --steps N gpurun_out/pmc_f/.../f_counter_collection.csv gpurun_out/pmc_w/.../w_counter_collection.csv \
            gpurun_out/pmc_f/.../f_kernel_trace.csv --workload dim256_f32_d4_b32_t0_bf16 > profiles/rNN_pmc_summary.json
`--steps N` = number of training steps the profiled process executed in total (bench.py: 2-3 warm-up + S timed + 3 of its
roofline pass); per-step figures divide by it.  Kernels that do not belong to the step (generator, predict) are kept out by
running bench.py with --no-fit --no-aux.
Corrections: FETCH_SIZE and WRITE_SIZE count KiB; gfx950 tallies a 128-byte read request as 64 bytes, so FETCH x 2.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def base_name(name):
    name = re.sub(r'^void\s+', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'<.*$', '', name)
    return name.replace('rvip::', '').strip()


def csrc_sha16():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'cmr-landmark-detection_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'cmr-landmark-detection_amd', 'csrc', '*.h'))
                    + [os.path.join(ROOT, 'include', 'rvip_hip.h')]):
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def counter_sums(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        f = base_name(r['Kernel_Name'])
        tot[f] += float(r['Counter_Value'])
        cnt[f] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv'); ap.add_argument('write_csv'); ap.add_argument('trace_csv')
    ap.add_argument('--steps', type=int, required=True)
    ap.add_argument('--workload', default='dim256_f32_d4_b32_t0_bf16')
    ap.add_argument('--only', default=None, help='regex: keep only kernels whose base name matches (default: all of this library)')
    a = ap.parse_args()
    ft, fc = counter_sums(a.fetch_csv, 'FETCH_SIZE')
    wt, wc = counter_sums(a.write_csv, 'WRITE_SIZE')
    dur, dn = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(a.trace_csv)):
        f = base_name(r['Kernel_Name'])
        dur[f] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        dn[f] += 1
    rx = re.compile(a.only) if a.only else None
    out = {'source': 'rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} (separate passes) -- python3 bench.py --no-graph --no-fit --no-aux --no-cpu-baseline; MI355X',
           'correction': 'FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as is; '
                         'counters are in KiB; durations are those of the (profiled) FETCH pass',
           'csrc_sha16': csrc_sha16(), 'workload': a.workload, 'steps_profiled': a.steps, 'kernels': {}}
    step_bytes = 0.0
    for f in sorted(ft, key=lambda k: -(ft[k] + wt.get(k, 0.0))):
        if not fc[f] or not wc.get(f) or (rx and not rx.search(f)):
            continue
        if f.startswith('at::') or 'elementwise' in f or 'Memcpy' in f or f.startswith('__amd'):      # torch plumbing (input upload), not the step
            continue
        fb = 2.0 * 1024.0 * ft[f] / fc[f]
        wb = 1024.0 * wt[f] / wc[f]
        us = dur[f] / max(dn[f], 1)
        per_step = fc[f] / float(a.steps)
        step_bytes += (fb + wb) * per_step
        out['kernels'][f] = dict(launches_sampled=fc[f], launches_per_step=round(per_step, 3), hbm_bytes_per_launch=round(fb + wb),
                                 fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb), avg_launch_us=round(us, 2),
                                 hbm_gb_per_s=round((fb + wb) / (us * 1e-6) / 1e9, 1) if us > 0 else None,
                                 hbm_bytes_per_step=round((fb + wb) * per_step))
    out['hbm_bytes_per_step'] = round(step_bytes)
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
