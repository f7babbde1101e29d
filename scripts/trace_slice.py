"""Diagnostic: print a slice of a rocprofv3 kernel trace as a per-queue timeline.
usage: python scripts/trace_slice.py trace.csv marker_substring [occurrence_from_end] [n_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
i0 = idx[-int(sys.argv[3]) if len(sys.argv) > 3 else -1]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + n]:
    name = r['Kernel_Name'].replace('void ', '')[:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r['Queue_Id']}  {name}")
