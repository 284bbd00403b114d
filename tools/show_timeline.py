#!/usr/bin/env python3
"""kernel timeline of the last iterations of a rocprofv3 kernel trace: python tools/show_timeline.py gpurun_out/<tag>_prof [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = [r for r in csv.DictReader(open(f)) if 'kmg' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n - 6:-6]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    nm = r['Kernel_Name'].split('(')[0].replace('void kmg::', '')[:30]
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{nm:32s} {s:9.1f} {e:9.1f} dur {e - s:7.1f} q{r.get('Queue_Id')} wg {r.get('Workgroup_Size_X')} grid {r.get('Grid_Size_X')} vgpr {r.get('VGPR_Count')} lds {r.get('LDS_Block_Size')}")
