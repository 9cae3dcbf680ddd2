#!/bin/bash
# ab6.sh NAME...: whole optimizer step of the ConvexDiffeomorphismNet and PathConnectedNet fits for each variant library
for n in "$@"; do echo "== $n"; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench_cdn.py 2>&1 | grep "us per"; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench_pcn.py --steps 300 2>&1 | grep "us per"; done
