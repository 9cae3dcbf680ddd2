set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/r04c/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r04c/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/kbench_wide.py > gpurun_out/r04c/kbench_wide.txt 2>&1 && cat gpurun_out/r04c/kbench_wide.txt
ROUND=r04c WIDE_SHAPES="256x1 350x3 512x2" bash tools/profile_round.sh wide c > gpurun_out/r04c/prof.log 2>&1; tail -3 gpurun_out/r04c/prof.log
