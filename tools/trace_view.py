#!/usr/bin/env python3
"""print the tail of a rocprofv3 kernel trace as a timeline: queue, kernel, start (us), duration (us)
   usage: tools/trace_view.py <kernel_trace.csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'rocclr' not in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = int(rows[-n]['Start_Timestamp'])
for r in rows[-n:]:
    name = r['Kernel_Name'].replace('void ', '').replace('dvs::', '')[:22]
    print(r['Queue_Id'].rjust(2), name.ljust(22), f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}")
