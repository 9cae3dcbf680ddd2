#!/bin/bash
# rocprofv3 kernel stats of the two PathConnectedNet fits (tools/kbench_pcn.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/exp_pcn
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o pcn -- python3 tools/kbench_pcn.py > gpurun_out/exp_pcn.log 2>&1 || exit 1
db=$(find $O -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" gpurun_out/exp_pcn.csv
rm -rf $O
grep "rnvp_\|icnn2\|pcn_update" gpurun_out/exp_pcn.csv | cut -c1-170
grep "checksum\|PCN fit" gpurun_out/exp_pcn.log
