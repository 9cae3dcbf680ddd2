#!/bin/bash
# rocprofv3 kernel stats of the layer-by-layer path (tools/kbench_wide.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/exp_wide
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o w -- python3 tools/kbench_wide.py $@ > gpurun_out/exp_wide.log 2>&1 || exit 1
db=$(find $O -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" gpurun_out/exp_wide.csv
rm -rf $O
head -25 gpurun_out/exp_wide.csv | cut -c1-190
tail -n 5 gpurun_out/exp_wide.log
