"""Summarise rocprofv3 PMC passes into profiles/<name>.json (HBM bytes per launch, per kernel family).

Usage (on the GPU box, each pass its own run, --kernel-trace only; see MI355X_MICROARCH.md section HBM):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py ... (same)
    python3 tools/pmc_summary.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv \
            gpurun_out/pmc_f/f_kernel_trace.csv > profiles/rNN_pmc_summary.json
Corrections: FETCH_SIZE and WRITE_SIZE count KiB; gfx950 tallies a 128-byte read request as 64 bytes, so FETCH x 2.
"""
import csv
import json
import re
import sys
from collections import defaultdict

FAMILIES = [  # (family key, regex on the demangled kernel name)
    ('conv3x3_igemm_ws', r'conv3x3_igemm_ws<'), ('conv3x3_igemm_dma', r'conv3x3_igemm_dma<'), ('conv3x3_igemm_v1', r'conv3x3_igemm<'),
    ('wgrad3x3_ws', r'wgrad3x3_ws<'), ('wgrad3x3_dma', r'wgrad3x3_dma<'), ('wgrad_fold', r'wgrad_fold_kernel'), ('fold_batch', r'fold_batch_'), ('pack_all', r'pack_all_kernel'),
    ('conv3x3_c1', r'conv3x3_c1|conv3d_c1'), ('c1_wgrad', r'c1_wgrad'), ('bn_stats', r'bn_stats_kernel'), ('bn_apply', r'bn_apply_kernel'),
    ('bn_bwd_reduce', r'bn_bwd_reduce_kernel'), ('bn_bwd_apply', r'bn_bwd_apply_kernel'), ('maxpool_bwd', r'maxpool_bwd_kernel'),
    ('upsample', r'upsample_kernel'), ('head_fwd', r'head_fwd'), ('head_bwd', r'head_bwd_kernel'), ('adam', r'adam_kernel'),
    ('fold_finalize', r'fold_finalize'),
]


def family(name):
    for key, rx in FAMILIES:
        if re.search(rx, name):
            return key
    return None


def counter_sums(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        f = family(r['Kernel_Name'])
        if f:
            tot[f] += float(r['Counter_Value'])
            cnt[f] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, trace_csv = sys.argv[1:4]
    ft, fc = counter_sums(fetch_csv, 'FETCH_SIZE')
    wt, wc = counter_sums(write_csv, 'WRITE_SIZE')
    dur, dn = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(trace_csv)):
        f = family(r['Kernel_Name'])
        if f:
            dur[f] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
            dn[f] += 1
    out = {'source': 'rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} (separate passes) -- python3 bench.py --steps 2 --warmup 1 '
                     '--no-cpu-baseline --no-graph; MI355X',
           'correction': 'FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as is; '
                         'counters are in KiB; durations are those of the (profiled) FETCH pass',
           'kernels': {}}
    for f in ft:
        if not fc[f] or not wc.get(f):
            continue
        fb = 2.0 * 1024.0 * ft[f] / fc[f]
        wb = 1024.0 * wt[f] / wc[f]
        us = dur[f] / max(dn[f], 1)
        out['kernels'][f] = dict(launches_sampled=fc[f], hbm_bytes_per_launch=round(fb + wb), fetch_bytes_per_launch=round(fb),
                                 write_bytes_per_launch=round(wb), avg_launch_us=round(us, 2),
                                 hbm_gb_per_s=round((fb + wb) / (us * 1e-6) / 1e9, 1) if us > 0 else None)
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
