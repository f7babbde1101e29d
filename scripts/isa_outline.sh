#!/bin/bash
# run-length outline of an ISA listing: where the scratch traffic, barriers, MFMA blocks and loads sit
grep -n "scratch_store\|scratch_load\|s_barrier\|^.LBB\|v_mfma\|global_load\|ds_read\|ds_write\|global_store" "$1" | awk '{print $1, $2}' \
 | awk '{k=$2; if (k==prev) {cnt++} else { if (prev!="") print first, prev, cnt; first=$1; prev=k; cnt=1 } } END {print first, prev, cnt}'
