#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc counter_collection.csv files (averages over the last launches)."""
import collections
import csv
import glob
import sys

KEYS = ['field_fwd_kernel<2, true>', 'field_fwd_kernel<2, false>', 'field_bwd_kernel', 'scatter_kernel', 'adam_kernel',
        'composite_fwd_kernel', 'composite_bwd_kernel', 'losses_kernel', 'ray_sampling_kernel', 'raygen_kernel']


def main(dirs):
    for d in dirs:
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                for k in KEYS:
                    if k in row['Kernel_Name']:
                        acc[k][row['Counter_Name']].append((float(row['Counter_Value']),
                                                            int(row['End_Timestamp']) - int(row['Start_Timestamp'])))
            print('==', f)
            for k in KEYS:
                for c, vals in acc.get(k, {}).items():
                    big = sorted(vals, key=lambda v: -v[0])[:3] if k == 'adam_kernel' else vals[-3:]
                    v = sum(x[0] for x in big) / len(big)
                    t = sum(x[1] for x in big) / len(big)
                    print('%-30s %-14s launches=%3d  avg=%.5g  avg_ns=%.0f' % (k, c, len(vals), v, t))


if __name__ == '__main__':
    main(sys.argv[1:])
