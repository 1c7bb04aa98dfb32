#!/usr/bin/env python3
"""Average every counter of rocprofv3 --pmc passes per kernel (regex on the kernel name):

    python tools/pmc_kernel.py 'field_bwd_kernel|field_fwd_kernel' DIR [DIR ...]
"""
import collections
import csv
import glob
import re
import sys


def main():
    pat = re.compile(sys.argv[1])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[2:]:
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                m = pat.search(row['Kernel_Name'])
                if m:
                    name = re.sub(r'\(anonymous namespace\)::|void ', '', row['Kernel_Name']).split('(')[0]
                    acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
                    acc[name]['_ns'].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    for name, cs in sorted(acc.items()):
        print(name)
        for c, v in sorted(cs.items()):
            print('   %-32s %16.1f   (%d samples)' % (c, sum(v) / len(v), len(v)))


if __name__ == '__main__':
    main()
