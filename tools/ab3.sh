#!/bin/bash
# ab3.sh NAME...: the bench's own timing (step kernel in sequence, update kernel, whole fit) for each variant library (on the GPU box)
for n in "$@"; do echo "== $n"; for k in 1 2; do INRFIT_LIB=$PWD/variants/libinrfit_$n.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --throughput-images 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('fit ms', d['ms_per_step'], 'step in-seq', r['kernel_us'], 'b2b', r['kernel_us_back_to_back'], 'update', r['update_kernel_us'], 'frac', r['frac'], d['fit_checksum'])"; done; done
