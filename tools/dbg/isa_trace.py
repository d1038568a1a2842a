#!/usr/bin/env python3
"""Condensed trace of one kernel of a hipcc -save-temps .s file: waits, barriers, branches, scratch traffic and global
accesses line by line, MFMA / ds_read / LDS-DMA runs as counts.  usage: isa_trace.py file.s mangled_kernel_name"""
import sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ':')
j = s.index('.end_amdhsa_kernel', i)
out, cnt = [], {}
def flush():
    global cnt
    if cnt:
        out.append('    ' + ' '.join(f'{k}x{v}' for k, v in cnt.items()))
        cnt = {}
for ln, l in enumerate(s[i:j].split('\n')):
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): cnt['mfma'] = cnt.get('mfma', 0) + 1
    elif op.startswith('ds_read'): cnt['dsr'] = cnt.get('dsr', 0) + 1
    elif op.startswith('global_load_lds'): cnt['dma'] = cnt.get('dma', 0) + 1
    elif op.startswith(('s_waitcnt', 's_barrier', 'scratch', 's_cbranch', 's_branch', 'global_', 's_setprio')) or t.endswith(':'):
        flush(); out.append(f'{ln}: {t}')
flush()
print('\n'.join(out))
