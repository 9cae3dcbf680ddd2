import re, collections, sys
path=sys.argv[1]; lo=int(sys.argv[2]); hi=int(sys.argv[3])
lines=[re.sub(r'\s+//.*','',l.strip()) for l in open(path).read().split('\n') if re.match(r'\s+\S',l)]
ops=[l.split()[0] for l in lines]
def cls(o,l):
    if o.startswith('v_mfma'): return 'MFMA'
    if o.startswith('ds_read') : return 'LDS read'
    if o.startswith('ds_write'): return 'LDS write'
    if o.startswith('ds_'): return 'LDS other'
    if o.startswith('global_') or o.startswith('buffer_'): return 'VMEM'
    if o=='s_waitcnt': return 's_waitcnt'
    if o=='s_nop': return 's_nop'
    if o=='s_barrier': return 's_barrier'
    if o.startswith('s_cbranch') or o.startswith('s_branch'): return 'branch'
    if o.startswith('s_'): return 'SALU'
    if o.startswith('v_accvgpr'): return 'v_accvgpr_read/write'
    if o in('v_mov_b32_e32','v_mov_b32_e64'): return 'v_mov'
    if o.startswith('v_exp') or o.startswith('v_log') or o.startswith('v_rcp') or o.startswith('v_sqrt'): return 'VALU trans'
    if 'permlane' in o or 'dpp' in l: return 'VALU dpp/permlane'
    if o.startswith('v_cmp'): return 'v_cmp'
    if o.startswith('v_cndmask'): return 'v_cndmask'
    if o.startswith('v_'): return 'VALU other'
    return 'other'
mf=[i for i in range(lo,hi) if ops[i].startswith('v_mfma')]
print("body",lo,hi,hi-lo,"mfma",len(mf))
tot=collections.Counter(cls(o,l) for o,l in zip(ops[lo:hi],lines[lo:hi]))
for k,v in tot.most_common(): print(f"  {k:24s} {v}")
names=[('top+fwd',0,272),('bwd',272,544),('mask+dL0',544,584),('dW',584,len(mf))]
prev=lo
for nm,a,b in names:
    h=mf[b-1]+1
    c=collections.Counter(cls(o,l) for o,l in zip(ops[prev:h],lines[prev:h]))
    print(nm, prev,h, dict(c), 'non-mfma/mfma %.2f'%((h-prev-c['MFMA'])/c['MFMA']))
    prev=h
print('tail',collections.Counter(cls(o,l) for o,l in zip(ops[prev:hi],lines[prev:hi])))
hv=collections.Counter(o for o,l in zip(ops[lo:hi],lines[lo:hi]) if cls(o,l)=='VALU other'); print(hv.most_common(20))
hs=collections.Counter(o for o,l in zip(ops[lo:hi],lines[lo:hi]) if cls(o,l)=='SALU'); print(hs.most_common(12))
