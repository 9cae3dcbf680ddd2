#!/bin/bash
# configs[3] (tools/kbench_pcn.py --case xyt): Q points per lane in the RealNVP forward (INR_RNVP_QF) / backward over points (INR_RNVP_QB)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for cfg in ${CFGS:-"1 1" "2 1" "4 1" "1 2" "2 2"}; do
  set -- $cfg
  export INR_RNVP_QF=$1 INR_RNVP_QB=$2
  O=gpurun_out/exp_q3_$1$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o pcn -- python3 tools/kbench_pcn.py --case xyt > gpurun_out/exp_q3_$1$2.log 2>&1 || exit 1
  db=$(find $O -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" gpurun_out/exp_q3_$1$2.csv
  rm -rf $O
  echo "== QF=$1 QB=$2"
  grep "rnvp_fwd\|rnvp_bwd_points\|pcn_update" gpurun_out/exp_q3_$1$2.csv | cut -c28-60,100-160
  grep "checksum\|PCN fit" gpurun_out/exp_q3_$1$2.log
done
