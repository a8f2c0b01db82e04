"""Timeline of a rocprofv3 kernel trace (csv): kernels over 0.8 ms and gaps over 2 ms, in launch order."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('cga::(anonymous namespace)::', '').split('(')[0]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d, gap = (e - s) / 1e6, (s - prev_end) / 1e6
    if d > lo or gap > 2:
        print(f"{(s - t0) / 1e6:9.2f} +{d:7.2f} gap {gap:7.2f}  {n[:60]}")
    prev_end = max(prev_end, e)
