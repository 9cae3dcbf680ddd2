#!/bin/bash
# ab2.sh NAME...: step kernel back to back and the whole optimizer step (fit) for each variant library (on the GPU box)
for n in "$@"; do echo "== $n"; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench.py --sizes 256 --rounds 5 --fit-steps 400 2>&1 | grep -v amdgpu.ids; INRFIT_LIB=$PWD/variants/libinrfit_$n.so python tools/kbench.py --sizes 256 --rounds 5 --fit-steps 400 2>&1 | grep -v amdgpu.ids; done
