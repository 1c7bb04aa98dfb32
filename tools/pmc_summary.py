#!/usr/bin/env python3
"""Per-entry-point HBM traffic from rocprofv3 --pmc passes (one pass FETCH_SIZE, one pass WRITE_SIZE).

    python tools/pmc_summary.py --steps 4 --out profiles/traffic.json DIR_FETCH DIR_WRITE

Counters are summed over every launch of the kernels behind an entry point and divided by the number of training
steps of the profiled run (bytes per STEP; bench.py divides by launches per step for the per-launch figure).
Corrections, as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE counts wide (16 B/lane) coalesced streaming reads at one half -> doubled for the Adam kernel, whose reads
are exactly that; WRITE_SIZE is exact for 16 B/lane streaming stores and for float atomics.  The 8-byte random
gathers of the field kernels are an uncalibrated access width: their FETCH_SIZE is reported uncorrected.
"""
import argparse
import collections
import csv
import glob
import json
import re

ENTRY = [
    (r'field_fwd_kernel<\d, true, false(, false)?>', 'nerf_field_fwd'),
    (r'field_fwd_kernel<\d, false, false(, false)?>', 'nerf_density_fwd'),
    (r'field_fwd_kernel<\d, true, false, true>', 'nerf_field_fwd_f16'),
    (r'field_fwd_kernel<\d, false, false, true>', 'nerf_density_fwd_f16'),
    (r'field_fwd_kernel<\d, true, true(, false)?>', 'nerf_field_from_features'),
    (r'field_fwd_kernel<\d, false, true(, false)?>', 'nerf_density_from_features'),
    (r'field_bwd_kernel', 'nerf_field_bwd'),
    (r'scatter_(small_)?kernel', 'nerf_hashgrid_scatter'),
    (r'encode_levels_kernel', 'nerf_hashgrid_encode_levels'),
    (r'adam_kernel', 'nerf_adam_step'),
    (r'table_update_kernel', 'nerf_table_update_binned'),
    (r'bin_(count|scan|fill)(_sorted)?_kernel', 'nerf_table_bin_levels'),
    (r'composite_loss_(fwd|finish)_kernel', 'nerf_composite_loss_fwd'),
    (r'composite_loss_bwd_kernel', 'nerf_composite_loss_bwd'),
    (r'sample_batch_kernel', 'nerf_sample_batch'),
    (r'composite_fwd_kernel', 'nerf_composite_fwd'),
    (r'composite_bwd_kernel', 'nerf_composite_bwd'),
    (r'losses_(sorted_)?kernel', 'nerf_losses'),
    (r'ray_sampling_kernel', 'nerf_coarse_to_fine'),
    (r'raygen_kernel', 'nerf_raygen'),
]
FETCH_X2 = {'nerf_adam_step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, required=True, help='training steps executed by the profiled command')
    ap.add_argument('--out', default=None)
    ap.add_argument('--build', default='?', help='tag of the profiled build, recorded in the output')
    ap.add_argument('--adam-grid', type=int, default=4194304, help='Grid_Size of the Adam launch on one hashed level')
    ap.add_argument('--levels-per-chunk', type=int, default=15, help='single-level scatter_kernel launches per backward chunk')
    ap.add_argument('--hashed-per-chunk', type=int, default=11, help='of which hashed levels (issued first)')
    ap.add_argument('--binned-item-bytes', type=float, default=15 * 8 * 524288 * 10.0,
                    help='bytes of corner-update items the binned update kernel reads per launch (2- and 8-byte loads, '
                         'counted in full by FETCH_SIZE, unlike its 16-byte p/m/v stream which is counted at one half)')
    ap.add_argument('dirs', nargs='+')
    a = ap.parse_args()
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    ns = collections.defaultdict(float)
    for d in a.dirs:
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                for pat, entry in ENTRY:
                    if re.search(pat, row['Kernel_Name']):
                        c = row['Counter_Name']
                        tot[entry][c] += float(row['Counter_Value'])
                        cnt[entry][c] += 1
                        ns[entry] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / len(a.dirs)
                        break
    # per-launch figures for the launch classes bench.py times separately (same names as nerf_amd._lib._launch_class)
    per_launch = collections.defaultdict(lambda: collections.defaultdict(list))
    seq = collections.defaultdict(int)
    for d in a.dirs:
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                kn, grid = row['Kernel_Name'], int(row['Grid_Size'])
                cls = None
                if re.search(r'scatter_kernel<', kn):
                    # host issue order per backward chunk: the hashed levels one by one, then the big dense levels one
                    # by one (nerf_amd/optim.py TablePipeline order 'A'); the small dense levels use scatter_small_kernel
                    key = (f, row['Counter_Name'])
                    pos = seq[key] % a.levels_per_chunk
                    seq[key] += 1
                    cls = 'nerf_hashgrid_scatter[1 hashed level]' if pos < a.hashed_per_chunk else 'nerf_hashgrid_scatter[1 dense level]'
                elif re.search(r'adam_kernel', kn) and grid == a.adam_grid:
                    cls = 'adam_table_slice'
                elif re.search(r'table_update_kernel', kn):
                    cls = 'nerf_table_update_binned'
                if cls:
                    per_launch[cls][row['Counter_Name']].append(float(row['Counter_Value']) * 1024)
    pl = {}
    for cls, c in per_launch.items():
        fetch = sum(c.get('FETCH_SIZE', [0])) / max(1, len(c.get('FETCH_SIZE', [0])))
        write = sum(c.get('WRITE_SIZE', [0])) / max(1, len(c.get('WRITE_SIZE', [0])))
        if cls == 'adam_table_slice':
            fetch *= 2
            cls = 'nerf_adam_step[table slice]'      # one hashed level: 2^T entries x F floats
        if cls == 'nerf_table_update_binned':        # wide p/m/v stream counted at one half, the item reads in full
            fetch = 2 * max(0.0, fetch - a.binned_item_bytes) + min(fetch, a.binned_item_bytes)
        pl[cls] = {'hbm_bytes': fetch + write, 'fetch_bytes': fetch, 'write_bytes': write}
    out = {'per_launch': pl, 'build': a.build}
    for entry in tot:
        fetch = tot[entry].get('FETCH_SIZE', 0.0) * 1024 / a.steps * (2 if entry in FETCH_X2 else 1)
        write = tot[entry].get('WRITE_SIZE', 0.0) * 1024 / a.steps
        launches = max(cnt[entry].values()) / a.steps
        out[entry] = {'hbm_bytes_per_step': fetch + write, 'fetch_bytes_per_step': fetch, 'write_bytes_per_step': write,
                      'launches_per_step': launches, 'serialized_ms_per_step': ns[entry] / a.steps / 1e6,
                      'fetch_doubled': entry in FETCH_X2}
    for k, v in sorted(pl.items()):
        print('%-52s %8.1f MB/launch (fetch %7.1f, write %7.1f)' % (k, v['hbm_bytes'] / 1e6, v['fetch_bytes'] / 1e6,
                                                                   v['write_bytes'] / 1e6))
    for k, v in sorted(((k, v) for k, v in out.items() if k not in ('per_launch', 'build')), key=lambda kv: -kv[1]['hbm_bytes_per_step']):
        print('%-28s %8.3f GB/step  (fetch %7.3f, write %7.3f)  %5.1f launches/step  %7.3f ms/step serialized' % (
            k, v['hbm_bytes_per_step'] / 1e9, v['fetch_bytes_per_step'] / 1e9, v['write_bytes_per_step'] / 1e9,
            v['launches_per_step'], v['serialized_ms_per_step']))
    if a.out:
        json.dump(out, open(a.out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
