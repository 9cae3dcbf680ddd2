#!/bin/bash
# ab5.sh NAME...: whole optimizer step of the PathConnectedNet fits (tools/kbench_pcn.py) for each variant library
for n in "$@"; do echo "== $n"; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench_pcn.py --steps 300 2>&1 | grep "us per"; done
