#!/bin/bash
# rocprofv3 kernel stats of the PathConnectedNet fits (tools/kbench_pcn.py --case $CASE); EXPU_LIST: values of $EXPVAR (default INR_RNVP_SHAPE; 0 = unset)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
VAR=${EXPVAR:-INR_RNVP_SHAPE}
for u in ${EXPU_LIST:-0}; do
  if [ "$u" = "0" ]; then unset $VAR; else export $VAR=$u; fi
  O=gpurun_out/exp_pcn_$u
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o pcn -- python3 tools/kbench_pcn.py --case ${CASE:-both} > gpurun_out/exp_pcn_$u.log 2>&1 || exit 1
  db=$(find $O -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" gpurun_out/exp_pcn_$u.csv
  rm -rf $O
  grep "rnvp_fwd\|rnvp_bwd\|pcn_update" gpurun_out/exp_pcn_$u.csv | cut -c1-170
  grep "checksum\|PCN fit" gpurun_out/exp_pcn_$u.log
done
