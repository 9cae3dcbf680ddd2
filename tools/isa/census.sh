#!/bin/bash
# usage: census.sh lib.so
cd /tmp/dis && L=/opt/rocm/lib/llvm/bin && $L/llvm-objcopy -O binary --only-section=.hip_fatbin $1 fatbin && $L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fatbin --output=k.co && $L/llvm-objdump -d --no-show-raw-insn k.co > k.s; n=$(grep -n "icnn_step_kernelILi130ELi2ELb1ELb0ELi0.*>:" k.s | cut -d: -f1); awk -v n=$n 'NR>=n' k.s | awk '/^$/{if(c++>0)exit} {print}' > step.s
python3 - <<'PY'
import re,subprocess
lines=[re.sub(r'\s+//.*','',l.strip()) for l in open('/tmp/dis/step.s').read().split('\n') if re.match(r'\s+\S',l)]
ops=[l.split()[0] for l in lines]
mf=[i for i,o in enumerate(ops) if o.startswith('v_mfma')]
bars=[i for i,o in enumerate(ops) if o=='s_barrier']
# loop body: from ~25 instrs before first mfma to the barrier after mfma #879
lo=mf[0]-26; hi=[b for b in bars if b>mf[879]][0]+6
print(subprocess.run(['python3','/tmp/census.py','/tmp/dis/step.s',str(lo),str(hi)],capture_output=True,text=True).stdout)
PY
