#!/bin/bash
# usage (GPU box): bash scripts/exp_wide.sh [B] [D] — scripts/time_wide.py under rocprofv3 for the product and every build_diag/lib_*.so
R=$GRAFT_REPO_ROOT
bash scripts/prof_any.sh prod python3 scripts/time_wide.py "$@" | head -5 | cut -c1-110
for lib in $R/build_diag/lib_*.so; do
  t=$(basename $lib .so); t=${t#lib_}
  MAPPO_HIP_LIB=$lib bash scripts/prof_any.sh $t python3 scripts/time_wide.py "$@" | sed -n 3,5p | sed "s/^/$t: /" | cut -c1-110
done
