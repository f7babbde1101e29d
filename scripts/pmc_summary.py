"""Compact summary of rocprofv3 counter-collection / kernel-trace CSVs under a directory: per (kernel, counter) the mean
value per dispatch and the mean dispatch duration.  usage: python scripts/pmc_summary.py DIR [name-filter]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "mlp_update"
csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: [0.0, 0, 0.0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if flt not in name and not (("FETCH" in r["Counter_Name"] or "WRITE" in r["Counter_Name"]) and "elementwise" in name and "copy" in name.lower()):
                continue
            short = name.split("(")[0].split("<")[0].split()[-1]
            k = (short, r["Counter_Name"], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
            a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
            if "Start_Timestamp" in r: a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc):
    v, n, d = acc[k]
    print(f"{k[0][:28]:28s} v={k[2]:4s} {k[1]:28s} {v / n:16.1f} /dispatch  n={n} ns={d / n:9.0f}")
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if flt in r["Name"] or float(r.get("Percentage", 0) or 0) > 3.0:
                short = r["Name"].split("(")[0].split("<")[0].split()[-1][:40]
                print(f"[stats] {short:40s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):10.0f} total_ns={r['TotalDurationNs']:>12s} pct={r['Percentage']}")
