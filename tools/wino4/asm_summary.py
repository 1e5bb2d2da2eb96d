"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (scratch traffic, MFMA, LDS, VMEM, VALU) -- used to
see where a register-tight kernel spills and how its main loop is laid out.  usage: asm_summary.py file.s kernel_substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^[_A-Za-z0-9]+:', l) and key in l and not l.startswith('.L'):
        start = i
        break
assert start is not None, 'kernel not found'
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
blk, stats, order = 'entry', {}, []
def bump(k):
    stats.setdefault(blk, {}).setdefault(k, 0)
    stats[blk][k] += 1
for l in lines[start:end + 1]:
    s = l.strip()
    m = re.match(r'^(\.LBB[0-9_]+):', s)
    if m:
        blk = m.group(1)
        continue
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    if blk not in order:
        order.append(blk)
    op = s.split()[0]
    if op.startswith('scratch_load'): bump('scr_ld')
    elif op.startswith('scratch_store'): bump('scr_st')
    elif op.startswith('v_mfma'): bump('mfma')
    elif op.startswith('ds_read') or op.startswith('ds_load'): bump('ds_rd')
    elif op.startswith('ds_write') or op.startswith('ds_store'): bump('ds_wr')
    elif op.startswith('buffer_load') or op.startswith('global_load'): bump('vm_ld')
    elif op.startswith('buffer_store') or op.startswith('global_store'): bump('vm_st')
    elif op.startswith('v_accvgpr'): bump('acc_mov')
    elif op.startswith('v_'): bump('valu')
    elif op.startswith('s_waitcnt'): bump('wait')
    elif op.startswith('s_barrier'): bump('barrier')
    elif op.startswith('s_'): bump('salu')
    else: bump('other')
keys = ['mfma', 'valu', 'ds_rd', 'ds_wr', 'vm_ld', 'vm_st', 'scr_ld', 'scr_st', 'acc_mov', 'wait', 'barrier', 'salu']
print('%-12s' % 'block' + ''.join('%8s' % k for k in keys))
for b in order:
    st = stats.get(b, {})
    if sum(st.values()) < 8 and not st.get('scr_ld') and not st.get('scr_st'):
        continue
    print('%-12s' % b + ''.join('%8d' % st.get(k, 0) for k in keys))
