"""Diagnostic: outline of a kernel's ISA (waits, barriers, branches, instruction-class counts in between).
usage: python scripts/isa_outline.py file.s mangled_kernel_name [first_line]"""
import re, sys
s = open(sys.argv[1]).read()
kn = sys.argv[2] + ':'
i = s.index(kn); j = s.index('.end_amdhsa_kernel', i)
cnt = {}; out = []
def flush():
    global cnt
    if cnt: out.append('   ' + ' '.join(f'{k}:{v}' for k, v in sorted(cnt.items())))
    cnt = {}
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if re.match(r'\.LBB\d+_\d+:', t): flush(); out.append(t[:60])
        continue
    op = t.split()[0]
    if op.startswith('s_waitcnt') or op == 's_barrier' or op.startswith('s_cbranch') or op == 's_branch':
        flush(); out.append(t)
    else:
        k = ('mfma' if 'mfma' in op else 'gload' if op.startswith('global_load') else 'gstore' if op.startswith('global_store')
             else 'ds_r' if op.startswith('ds_read') else 'ds_w' if op.startswith('ds_write') else 'scratch' if op.startswith('scratch')
             else 'valu' if op.startswith('v_') else 'salu')
        cnt[k] = cnt.get(k, 0) + 1
flush()
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
print('\n'.join(out[first:]))
