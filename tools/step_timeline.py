"""One training step of the captured graph, launch by launch: position, start offset, duration, gap to the previous kernel.

    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-fit --no-aux --no-roofline-pass --steps 30 --warmup 5
    python3 tools/step_timeline.py DIR/.../t_kernel_trace.csv [labels.json] > timeline.txt

A step = the kernels from one convert_kernel (stage_input, the first launch of the step) to the next.  Durations are medians over
the traced steps of the replayed graph; `gap` = start minus the previous kernel's end.  labels.json (optional, `bench.py --dump-labels`)
holds the engine's launch labels in launch order and is joined by position."""
import csv
import json
import re
import statistics
import sys


def short(name):
    name = re.sub(r'^void\s+', '', name).replace('rvip::', '')
    name = re.sub(r'\(.*$', '', name)
    return name[:78]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    labels = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
    starts = [i for i, r in enumerate(rows) if 'convert_kernel' in r[2]]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    n = statistics.mode(len(s) for s in steps)
    steps = [s for s in steps if len(s) == n][-20:]
    print('# %d steps of %d launches; step = %.1f us (median, first start to last end)' % (
        len(steps), n, statistics.median((s[-1][1] - s[0][0]) * 1e-3 for s in steps)))
    print('# idx  start_us   dur_us  gap_us  kernel | label')
    tot_gap = 0.0
    for i in range(n):
        dur = statistics.median((s[i][1] - s[i][0]) * 1e-3 for s in steps)
        st = statistics.median((s[i][0] - s[0][0]) * 1e-3 for s in steps)
        gap = statistics.median((s[i][0] - s[i - 1][1]) * 1e-3 for s in steps) if i else 0.0
        tot_gap += gap
        lab = ''
        if labels and len(labels) == n:
            lab = ' | ' + labels[i]
        print('%4d %9.1f %8.1f %7.2f  %s%s' % (i, st, dur, gap, short(steps[0][i][2]), lab))
    print('# sum of gaps %.1f us' % tot_gap)


if __name__ == '__main__':
    main()
