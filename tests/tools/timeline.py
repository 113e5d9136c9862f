"""Timeline of the last build in a rocprofv3 kernel trace: python tests/tools/timeline.py <kernel_trace.csv> [min_us=300] [queues]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sa_symbols' in r['Kernel_Name'] or 'msd_hist_raw' in r['Kernel_Name']]
last = rows[idx[-1]:] if idx else rows
t0 = int(last[0]['Start_Timestamp'])
agg = {}
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].replace('pss::', '').replace('void ', '').split('(')[0]
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += d
    if d >= min_us:
        q = f" q={r['Queue_Id']}" if len(sys.argv) > 3 and 'Queue_Id' in r else ''
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.2f} ms {d:9.1f} us  {name[:50]:50s} grid={r['Grid_Size_X']} lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']}{q}")
print('--- per kernel (last build) ---')
tot = 0.0
for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += d
    if d >= 200:
        print(f'{name[:60]:60s} {c:4d} calls {d / 1e3:8.2f} ms')
end = max(int(r['End_Timestamp']) for r in last)
print(f'kernel time {tot / 1e3:.2f} ms, span {(end - t0) / 1e6:.2f} ms')
