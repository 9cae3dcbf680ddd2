#!/bin/bash
# tools/prof_stats.sh TAG -- <program and arguments>: rocprofv3 kernel trace of the program on the GPU box, summarised into
# gpurun_out/TAG_kernel_stats.csv (the raw trace is deleted).  The program follows `--` directly (no env / bash -c wrappers).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
tag=$1; shift; shift
d=gpurun_out/prof_$tag
rm -rf $d
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $d -o t -- "$@" > gpurun_out/${tag}_prof.log 2>&1 || { echo "rocprofv3 failed"; tail -5 gpurun_out/${tag}_prof.log; exit 1; }
db=$(find $d -name "*.db" | head -1)
if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" gpurun_out/${tag}_kernel_stats.csv; else cp "$(find $d -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv; fi
rm -rf $d
python3 - "$tag" <<'PY'
import csv, sys
rows = list(csv.reader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:13]:
    print(f"{r[0][:64]:64s} " + " ".join(f"{c:>12s}" for c in (r[1], r[3], r[5], r[6]) ))
PY
