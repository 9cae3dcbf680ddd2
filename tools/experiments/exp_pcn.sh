#!/bin/bash
# rocprofv3 kernel stats of the 256x256 PathConnectedNet fit (tools/kbench_pcn.py --case xy) per RealNVP launch shape (INR_RNVP_SHAPE = U)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for u in ${EXPU_LIST:-1 2 4}; do
  export INR_RNVP_SHAPE=$u
  O=gpurun_out/exp_pcn_$u
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o pcn -- python3 tools/kbench_pcn.py --case xy > gpurun_out/exp_pcn_$u.log 2>&1 || exit 1
  db=$(find $O -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" gpurun_out/exp_pcn_$u.csv
  rm -rf $O
  grep "rnvp_fwd\|rnvp_bwd\|pcn_update" gpurun_out/exp_pcn_$u.csv | cut -c1-170
  grep "checksum\|PCN fit" gpurun_out/exp_pcn_$u.log
done
