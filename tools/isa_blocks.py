"""Basic-block view of one kernel of a gfx950 .s file: per block the MFMA / v_exp / VALU / ds_read counts and the issue order
(M = MFMA, e = v_exp, . = other VALU, d = ds_read, B = buffer_load, | = s_barrier). Usage: isa_blocks.py file.s <mangled-substring>"""
import re
import sys

txt = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\S+):\s*; @', txt, re.M)
blocks = re.split(r'^_Z\S+:\s*; @.*$', txt, flags=re.M)
for n, b in zip(names, blocks[1:]):
    if sys.argv[2] not in n:
        continue
    cur, cnt, order = None, {}, []
    for line in b.splitlines():
        l = line.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            cur = m.group(1)
            order.append(cur)
            cnt[cur] = {'M': 0, 'e': 0, 'v': 0, 'd': 0, 'br': [], 'seq': ''}
            continue
        if cur is None:
            continue
        c = cnt[cur]
        if l.startswith('v_mfma'):
            c['M'] += 1; c['seq'] += 'M'
        elif l.startswith('v_exp'):
            c['e'] += 1; c['seq'] += 'e'
        elif l.startswith('ds_read'):
            c['d'] += 1; c['seq'] += 'd'
        elif l.startswith('v_'):
            c['v'] += 1; c['seq'] += '.'
        elif l.startswith('s_cbranch') or l.startswith('s_branch'):
            c['br'].append(l.split()[-1])
        elif l.startswith('s_barrier'):
            c['seq'] += '|'
        elif l.startswith('buffer_load'):
            c['seq'] += 'B'
        elif l.startswith('s_waitcnt'):
            c['seq'] += 'w'
    for k in order:
        c = cnt[k]
        if c['M'] or c['e']:
            print(k, 'M', c['M'], 'e', c['e'], 'v', c['v'], 'd', c['d'], c['br'])
            print('    ', c['seq'])
