#!/bin/bash
# usage (GPU box, via gpurun): bash scripts/pmc_hbm.sh   -> gpurun_out/dual_update_hbm_pmc.{txt,json}
# HBM traffic of the dominant kernel (mlp_update16_dual_kernel at BASELINE config-2 size): FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (counters only), corrected as MI355X_MICROARCH.md prescribes.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_hbm; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  d=$(echo $c | cut -c1-10 | tr " " _)
  PMC_CALIB=0 timeout -k 10 180 rocprofv3 --pmc $c --output-format csv -d $OUT/$d -- python3 $R/scripts/pmc_dual.py > $OUT/log_$d.txt 2>&1 || echo fail $c
done
cd $R
python3 scripts/pmc_summary.py $OUT mlp_update16 | cut -c1-160 > gpurun_out/dual_update_hbm_pmc.txt
cat gpurun_out/dual_update_hbm_pmc.txt
python3 - <<'PY'
import hashlib, json, os, re
def kernel_source_sha256():
    # the sources the dominant kernel is compiled from: a later edit makes the committed measurement stale (bench.py checks)
    h = hashlib.sha256()
    for f in ("mlp_upd16.h", "mlp_core.h", "common.h"):
        h.update(open(os.path.join("mappo_amd", "csrc", f), "rb").read())
    return h.hexdigest()
vals = {}
for line in open("gpurun_out/dual_update_hbm_pmc.txt"):
    t = line.split()
    if len(t) > 3 and t[0].startswith("mlp_update16_dual"):
        vals[t[2]] = float(t[3])
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
out = dict(kernel="mlp_update16_dual_kernel", samples=76800, source_sha256=kernel_source_sha256(), fetch_size_kb=fetch_kb, write_size_kb=write_kb,
           tcc_ea0_rdreq=vals.get("TCC_EA0_RDREQ_sum"), tcc_ea0_wrreq=vals.get("TCC_EA0_WRREQ_sum"),
           traffic_bytes=int((2 * fetch_kb + write_kb) * 1024),
           note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/pmc_dual.py (6 launches, mean per "
                "dispatch); traffic = 2 x FETCH_SIZE (gfx950: 128-B read requests are tallied at 64 B) + WRITE_SIZE.  The doubling "
                "over-corrects the dword loads of the loss inputs (3.7 MB), so this is an upper bound of the true HBM bytes")
json.dump(out, open("gpurun_out/dual_update_hbm_pmc.json", "w"), indent=1)
print(out)
PY
