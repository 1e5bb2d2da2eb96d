"""Compressed view of a kernel's MFMA loop in a hipcc -S listing: one letter per instruction (M mfma, v VALU, a accvgpr move,
R/W LDS read/write, B/G buffer/global load, S store, [..] waitcnt, |BAR|), runs of v / a / s counted.
usage: loopview.py file.s kernel_substring [lines_before] """
import re, sys
L = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 60
st = [i for i, l in enumerate(L) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ':' in l)][0]
en = next(i for i in range(st, len(L)) if L[i].strip().startswith('s_endpgm'))
K = L[st:en + 1]
idx = [i for i, l in enumerate(K) if 'v_mfma' in l]
def cat(s):
    op = s.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('v_accvgpr'): return 'a'
    if op.startswith('ds_read'): return 'R'
    if op.startswith('ds_write'): return 'W'
    if op.startswith('buffer_load'): return 'B'
    if op.startswith('global_load'): return 'G'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'S'
    if op.startswith('scratch_'): return '!'
    if op.startswith('s_waitcnt'): return '[' + s.split(None, 1)[1].replace(' ', '') + ']'
    if op.startswith('s_barrier'): return '|BAR|'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return '<br>'
    if op.startswith('s_nop'): return 'n'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_'): return 's'
    return '?'
out = []
for l in K[max(0, idx[0] - before): idx[-1] + 30]:
    s = l.strip()
    if not s or s.startswith(';'): continue
    m = re.match(r'^(\.LBB[0-9_]+):', s)
    if m:
        out.append('\n' + m.group(1) + ': ')
        continue
    if s.startswith('.'): continue
    out.append(cat(s))
txt = ''.join(out)
for ch in 'vasn':
    txt = re.sub(ch + r'{2,}', lambda m: '%s%d ' % (ch, len(m.group(0))), txt)
print('MFMAs', len(idx))
print(txt)
