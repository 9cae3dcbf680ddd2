#!/bin/bash
# ab4.sh NAME...: 64 images per launch (BASELINE configs[2] per-GPU share): whole-fit throughput for each variant library
for n in "$@"; do echo "== $n"; for k in 1 2; do INRFIT_LIB=$PWD/variants/libinrfit_$n.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --throughput-images 64 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['throughput_mode']
print('fits/s', t['fits_per_s'], 'us/step/image', t['us_per_optimizer_step_per_image'], 'miou', t['miou_vs_unaries'])"; done; done
