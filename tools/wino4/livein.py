"""VGPRs that are live into the MFMA loop of a kernel (read before written inside the loop body) and, of those, the ones never
written inside the loop (loop invariants kept in registers) -- from a hipcc -S listing.  usage: livein.py file.s kernel_substring"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
st = [i for i, l in enumerate(L) if l.startswith('_Z') and key in l][0]
en = next(i for i in range(st, len(L)) if L[i].strip().startswith('s_endpgm'))
K = L[st:en + 1]
idx = [i for i, l in enumerate(K) if 'v_mfma' in l]
# loop = from the label before the first mfma to the backward branch after the last mfma
lo = max(i for i in range(idx[0]) if re.match(r'^\.LBB', K[i].strip()) and 'Loop Header' in K[i] or (re.match(r'^\.LBB', K[i].strip()) and i < idx[0]))
hi = next(i for i in range(idx[-1], len(K)) if K[i].strip().startswith('s_cbranch'))
def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
written, livein, everw = set(), set(), set()
for l in K[lo:hi + 1]:
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.startswith('s_') : continue
    s = s.split(';')[0]
    parts = s.split(None, 1)
    if len(parts) < 2: continue
    op, rest = parts
    ops = [o.strip() for o in rest.split(',')]
    stores = op.startswith(('ds_write', 'global_store', 'buffer_store', 'scratch_store', 'v_cmp', 'v_cmpx'))
    if op.startswith('v_mfma'):
        dst, srcs = ops[0], ops[1:]
    elif stores:
        dst, srcs = '', ops
    else:
        dst, srcs = ops[0], ops[1:]
    for r in [x for o in srcs for x in regs(o)]:
        if r not in written: livein.add(r)
    for r in regs(dst):
        written.add(r); everw.add(r)
inv = sorted(r for r in livein if r not in everw)
print('loop lines %d..%d; live-in VGPRs %d; of them never written in the loop (invariants) %d' % (lo, hi, len(livein), len(inv)))
print('invariants:', inv)
