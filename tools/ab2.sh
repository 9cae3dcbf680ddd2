#!/bin/bash
# ab2.sh NAME...: kbench of the two-hidden-layer step kernel for each variant library (on the GPU box)
for n in "$@"; do echo "== $n"; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench.py --layers 2 --sizes 256 --rounds 5 2>&1 | grep -v amdgpu.ids; done
