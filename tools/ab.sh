#!/bin/bash
# ab.sh NAME...: run kbench for each variant library (on the GPU box)
for n in "$@"; do echo "== $n"; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench.py --sizes 256 512 --rounds 5 2>&1 | grep -v amdgpu.ids; done
