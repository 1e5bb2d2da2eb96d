"""Approximate VGPR pressure along the MFMA loop of a kernel (linear scan of the loop body, backwards liveness; inner branches
are treated as straight-line code).  usage: pressure.py file.s kernel_substring"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
st = [i for i, l in enumerate(L) if l.startswith('_Z') and key in l][0]
en = next(i for i in range(st, len(L)) if L[i].strip().startswith('s_endpgm'))
K = L[st:en + 1]
idx = [i for i, l in enumerate(K) if 'v_mfma' in l]
lo = max(i for i in range(idx[0]) if re.match(r'^\.LBB', K[i].strip()))
hi = next(i for i in range(idx[-1], len(K)) if K[i].strip().startswith('s_cbranch'))
def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
ins = []
nm = 0
for l in K[lo:hi + 1]:
    s = l.strip().split(';')[0].strip()
    if not s or s.startswith('.') or s.startswith('s_'): continue
    parts = s.split(None, 1)
    if len(parts) < 2: continue
    op, rest = parts
    ops = [o.strip() for o in rest.split(',')]
    stores = op.startswith(('ds_write', 'global_store', 'buffer_store', 'scratch_store', 'v_cmp', 'v_cmpx'))
    if 'v_mfma' in op: nm += 1
    if stores: d, srcs = [], [x for o in ops for x in regs(o)]
    else:
        d, srcs = regs(ops[0]), [x for o in ops[1:] for x in regs(o)]
        if op.startswith('v_mfma') or op.startswith('v_fmac') or op.startswith('v_mac'): srcs += d
    ins.append((op, d, srcs, nm))
# loop-carried: live at the end = live at the start (iterate twice)
live = set()
for _ in range(2):
    press = []
    for op, d, srcs, m in reversed(ins):
        live -= set(d)
        live |= set(srcs)
        press.append((len(live), m, op))
    press.reverse()
mx = max(press)
print('max live VGPRs %d at mfma #%d (%s)' % mx)
# pressure at each step boundary
last = -1
for p, m, op in press:
    if m // 4 != last and m % 4 == 0:
        last = m // 4
        print('step %2d: %d' % (last, p), end='   ')
print()
