#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of one gpurun round from gpurun_out/ into profiles/ and refreshes
profiles/traffic.json.   usage: tools/summarise_profiles.py r01c"""
import csv, glob, json, shutil, statistics, sys
tag = sys.argv[1]
shutil.copy(glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_stats.csv')[0], f'profiles/{tag}_kernel_stats.csv')
shutil.copy(f'gpurun_out/bench_{tag}.json', f'profiles/{tag}_bench.json')
rows, hdr = [], None
for f in glob.glob(f'gpurun_out/pmc_fetch_{tag}/*/*counter_collection.csv') + glob.glob(f'gpurun_out/pmc_write_{tag}/*/*counter_collection.csv'):
    r = list(csv.reader(open(f))); hdr = r[0]; rows += [x for x in r[1:] if 'kmg::' in x[8]]
with open(f'profiles/{tag}_pmc_fetch_write.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(hdr); w.writerows(rows)
ni, ci, vi = hdr.index('Kernel_Name'), hdr.index('Counter_Name'), hdr.index('Counter_Value')
def mean(kern, ctr):
    v = [float(x[vi]) for x in rows if kern in x[ni] and x[ci] == ctr]
    return statistics.mean(v) if v else None
kernels = {}
for kern in ['k_labels', 'k_cube', 'k_cell_candidates', 'k_reduce_partials', 'k_histogram', 'k_assign']:
    f, w = mean(kern, 'FETCH_SIZE'), mean(kern, 'WRITE_SIZE')
    if f is not None:
        kernels[kern] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
t = json.load(open('profiles/traffic.json'))
stream = 8192 * 8192 * 4
lab = kernels.get('k_labels')
if lab:
    gather = lab['FETCH_SIZE_KiB'] * 1024 - stream / 2
    t['bytes_per_launch']['k_labels'] = stream + gather + lab['WRITE_SIZE_KiB'] * 1024
    note = (f"k_labels: FETCH_SIZE {lab['FETCH_SIZE_KiB']*1024/1e6:.1f} MB raw.  The pixel stream (268.4 MB, wide coalesced) is tallied at half "
            f"(134.2 MB, gfx950 correction of MI355X_MICROARCH.md section HBM); the remaining {gather/1e6:.1f} MB are the line fills of the "
            "label-table gathers (sub-line accesses, at face value; served by the Infinity Cache, which FETCH_SIZE counts).  "
            f"traffic = 268.4 (pixels) + {gather/1e6:.1f} (table fills) + {lab['WRITE_SIZE_KiB']*1024/1e6:.1f} (labels written, WRITE_SIZE exact) MB "
            "vs 536.9 MB algorithmic.")
else:
    note = ""
t['rounds'][tag] = {"kernels": kernels, "note": note}
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
d = json.load(open(f'profiles/{tag}_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline'])
print({k: round(v['ms_per_launch'], 4) for k, v in d['kernels'].items()})
print(t['bytes_per_launch'])
for r in csv.DictReader(open(f'profiles/{tag}_kernel_stats.csv')):
    if 'kmg::' in r['Name']:
        print(f"  rocprof {r['Name'][:40]:40s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
