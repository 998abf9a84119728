#!/usr/bin/env python
"""Print ONE step of a rocprofv3 --kernel-trace (csv) as the ordered list of kernel dispatches with their durations and
the idle gap in front of each:   python tools/trace_step.py <dir with *_kernel_trace.csv> [step index from the end]
Used to read where a small-batch forward (ResNet-50 at batch 8, the detector at batch 1) spends its time, launch by launch."""
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if 'rocclr' not in r['Kernel_Name']]
    # a step starts at every nchw_to_nhwc4 / stem kernel (the detector's, the ResNet trunk's)
    starts = [i for i, r in enumerate(rows) if 'nchw_to_nhwc4' in r['Kernel_Name'] or 'stem_conv' in r['Kernel_Name'] or 'stem7x7' in r['Kernel_Name']]
    a, b = starts[-back - 1], starts[-back]
    t0 = int(rows[a]['Start_Timestamp'])
    prev_end = t0
    tot = 0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('void ', '').replace('xdet::', '').split('(')[0]
        print('%8.1f us  +%6.1f gap  %7.1f us  grid %-8s %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                                                               r.get('Grid_Size', r.get('Grid_Size_X', '?')), name[:90]))
        prev_end = max(prev_end, e)
        tot += e - s
    print('step: %.1f us wall, %.1f us of kernels, %d dispatches' % ((prev_end - t0) / 1e3, tot / 1e3, b - a))


if __name__ == '__main__':
    main()
