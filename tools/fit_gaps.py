"""What sits between consecutive training steps: bench loop against Model.fit.

    rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux --no-roofline-pass --steps 30 --warmup 5
    python3 tools/fit_gaps.py DIR/.../t_kernel_trace.csv [DIR/.../t_memory_copy_trace.csv]

A step = convert_kernel (first launch) .. pack_all_kernel (last).  For every pair of consecutive steps: the gap from the end of
pack_all to the start of the next convert, and the launches / copies recorded inside it.  Gaps are grouped by what they contain."""
import csv
import re
import statistics
import sys


def short(name):
    name = re.sub(r'^void\s+', '', name).replace('rvip::', '')
    return re.sub(r'[<(].*$', '', name)[:40]


def main():
    ev = []
    for r in csv.DictReader(open(sys.argv[1])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    if len(sys.argv) > 2:
        for r in csv.DictReader(open(sys.argv[2])):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy:' + r.get('Direction', r.get('Name', '?'))))
    ev.sort()
    ends = [i for i, e in enumerate(ev) if e[2].startswith('pack_all_kernel')]
    groups = {}
    for i in ends:
        j = i + 1
        inside = []
        while j < len(ev) and not ev[j][2].startswith('convert_kernel'):
            inside.append(ev[j])
            j += 1
            if len(inside) > 12:
                break
        if j >= len(ev) or len(inside) > 12:
            continue
        key = ' + '.join(e[2] for e in inside if not e[2].startswith('copy:MEMORY_COPY_HOST_TO_DEVICE')) or '(nothing)'
        g = groups.setdefault(key, [])
        busy = sum(min(e[1], ev[j][0]) - max(e[0], ev[i][1]) for e in inside if e[0] < ev[j][0] and e[1] > ev[i][1] and not e[2].startswith('copy:MEMORY_COPY_HOST_TO_DEVICE'))
        g.append(((ev[j][0] - ev[i][1]) * 1e-3, busy * 1e-3))
    for key, g in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        gaps = [x[0] for x in g]
        print('%4d step boundaries: gap median %7.1f us (p10 %7.1f, p90 %7.1f), busy inside %6.1f us | %s' % (
            len(g), statistics.median(gaps), sorted(gaps)[len(gaps) // 10], sorted(gaps)[(9 * len(gaps)) // 10],
            statistics.median(x[1] for x in g), key))


if __name__ == '__main__':
    main()
