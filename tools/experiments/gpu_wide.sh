set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
timeout -k 10 500 python -m pytest tests/test_gpu_icnn.py tests/test_gpu_teaser.py -q -x -k "wide or star" > gpurun_out/r04w/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r04w/tests.log
timeout -k 10 200 python tools/kbench_wide.py > gpurun_out/r04w/kbench_new.txt 2>&1 && cat gpurun_out/r04w/kbench_new.txt
for v in $WIDE_VARIANTS; do INRFIT_LIB=variants/libinrfit_$v.so timeout -k 10 200 python tools/kbench_wide.py 256x1 350x3 > gpurun_out/r04w/kbench_$v.txt 2>&1 && cat gpurun_out/r04w/kbench_$v.txt; done
ROUND=r04w WIDE_SHAPES="256x1 350x3" bash tools/profile_round.sh wide c > gpurun_out/r04w/prof.log 2>&1; tail -3 gpurun_out/r04w/prof.log
