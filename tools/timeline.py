#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: start / end of every dispatch relative to the step's
first kernel, with its stream (queue), so that what runs beside what -- and what the step's end waits for -- can be read.

    python tools/timeline.py DIR_WITH_kernel_trace.csv [--step 12]
"""
import argparse
import csv
import glob
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--step', type=int, default=-2, help='which raygen-delimited step to print (negative: from the end)')
    a = ap.parse_args()
    rows = []
    for f in glob.glob(a.dir + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '?'))))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if 'raygen_kernel' in r[2]]
    i0 = starts[a.step]
    i1 = starts[a.step + 1] if a.step + 1 != 0 and a.step + 1 < len(starts) else len(rows)
    t0 = rows[i0][0]
    short = lambda n: re.sub(r'\(anonymous namespace\)::|void |\(.*', '', n)[:60]
    for s, e, n, q in rows[i0:i1]:
        print('%9.1f %9.1f  %7.1f us  q%-4s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
    print('step span %.1f us' % ((rows[i1 - 1][1] - t0) / 1e3))


if __name__ == '__main__':
    main()
