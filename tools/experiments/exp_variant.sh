#!/bin/bash
# rocprofv3 kernel stats of tools/kbench_pcn.py with the library variants/libinrfit_$1.so (tools/build_variant.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
export INRFIT_LIB=$PWD/variants/libinrfit_$1.so
O=gpurun_out/exp_var_$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o pcn -- python3 tools/kbench_pcn.py > gpurun_out/exp_var_$1.log 2>&1 || exit 1
db=$(find $O -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" gpurun_out/exp_var_$1.csv
rm -rf $O
grep "rnvp_fwd\|rnvp_bwd" gpurun_out/exp_var_$1.csv | cut -c1-170
grep "checksum\|PCN fit" gpurun_out/exp_var_$1.log
