#!/bin/bash
# usage (GPU box, via gpurun): bash scripts/prof_configs.sh "c3 c4 c5" <tag>  -> gpurun_out/prof_<tag>_<cfg>.txt
# rocprofv3 --kernel-trace --stats of scripts/bench_configs.py for the other BASELINE configs; compact per-kernel summaries.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${2:-r02}
for c in $1; do
  OUT=$R/gpurun_out/prof_${TAG}_$c; rm -rf $OUT; mkdir -p $OUT; cd /tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/bench_configs.py $c > $OUT/bench.jsonl 2> $OUT/err.txt || echo "failed $c"
  cd $R
  python3 - "$OUT" "$c" <<'PY' > gpurun_out/prof_${TAG}_$c.txt
import csv, glob, os, sys
root, cfg = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
print("# rocprofv3 --kernel-trace --stats -- python3 scripts/bench_configs.py", cfg)
try:
    print("# bench line:", open(os.path.join(root, "bench.jsonl")).read().strip().splitlines()[-1][:600])
except Exception as e:
    print("# bench line missing:", e)
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f, newline="")))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print(f"{'kernel':52s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:22]:
        short = r["Name"].split("(")[0].split("<")[0].split()[-1][:52]
        print(f"{short:52s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")
PY
  cat gpurun_out/prof_${TAG}_$c.txt | cut -c1-110
  rm -rf $OUT
done
