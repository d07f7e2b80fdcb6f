"""per-kernel instruction / register statistics of a hipcc --save-temps .s file: python tools/isa_stats.py file.s [name-substring]"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
name = None
stats = {}
for l in lines:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name = m.group(1)
        stats[name] = dict(n=0, valu=0, salu=0, vmem=0, lds=0, wait=0, nop=0, dpp=0, vgpr=None, sgpr=None)
        continue
    if name is None:
        continue
    t = l.strip()
    st = stats[name]
    if t.startswith('.amdhsa_next_free_vgpr'):
        st['vgpr'] = t.split()[-1]
    elif t.startswith('.amdhsa_next_free_sgpr'):
        st['sgpr'] = t.split()[-1]
    elif re.match(r'(v_|s_|ds_|global_|buffer_|scratch_|flat_)', t):
        st['n'] += 1
        if t.startswith('v_'):
            st['valu'] += 1
            if 'dpp' in t or 'quad_perm' in t or 'row_' in t:
                st['dpp'] += 1
        elif t.startswith('s_waitcnt'):
            st['wait'] += 1
        elif t.startswith('s_nop'):
            st['nop'] += 1
        elif t.startswith('s_'):
            st['salu'] += 1
        elif t.startswith('ds_'):
            st['lds'] += 1
        else:
            st['vmem'] += 1
for k, v in stats.items():
    if pat in k and v['n']:
        print(k[:90], v)
