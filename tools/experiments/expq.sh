#!/bin/bash
# experiment: coupling-flow point kernels with U lanes per point and Q points per lane (INR_FLOW_SHAPE = 10 U + Q), rocprofv3 kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for q in ${EXPQ_LIST:-0}; do
  export INR_FLOW_SHAPE=$q
  O=gpurun_out/expq_$q
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cdn -- python3 tools/kbench_cdn.py > gpurun_out/expq_$q.log 2>&1 || exit 1
  db=$(find $O -name "*.db" | head -1)
  python3 tools/rocpd_stats.py "$db" gpurun_out/expq_$q.csv
  rm -rf $O
  grep "flow_fwd\|flow_bwd_points\|checksum\|CDN fit" gpurun_out/expq_$q.csv gpurun_out/expq_$q.log | cut -c1-200
done
