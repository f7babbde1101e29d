#!/bin/bash
# usage (GPU box): bash scripts/prof_any.sh <tag> python3 <script> [args]   -> gpurun_out/prof_<tag>.txt (compact per-kernel stats)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
SCRIPT=$2; shift 2
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$SCRIPT "$@" > $OUT/out.txt 2> $OUT/err.txt || echo "failed $TAG"
cd $R
python3 - "$OUT" "$TAG" <<'PY' > gpurun_out/prof_$TAG.txt
import csv, glob, os, sys
root, tag = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
print("#", tag, open(os.path.join(root, "out.txt")).read().strip()[-400:])
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f, newline="")))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print(f"{'kernel':52s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:8]:
        short = r["Name"].split("(")[0].split("<")[0].split()[-1][:52]
        print(f"{short:52s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")
PY
cut -c1-120 gpurun_out/prof_$TAG.txt
rm -rf $OUT
