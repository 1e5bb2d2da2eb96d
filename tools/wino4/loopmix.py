"""Instruction mix per basic block of ONE kernel in a hipcc -S listing, the whole function (up to .Lfunc_end), with the vector
integer adds (LDS address arithmetic) and packed fp32 instructions counted apart.  usage: loopmix.py file.s kernel_substring [min]"""
import collections
import re
import sys
path, key = sys.argv[1], sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^[_A-Za-z0-9]+:', l) and key in l and not l.startswith('.L'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blk, stats = 'entry', collections.OrderedDict()
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r'^(\.LBB[0-9_]+):', s)
    if m:
        blk = m.group(1)
        continue
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    op = s.split()[0]
    d = stats.setdefault(blk, collections.Counter())
    if op.startswith('v_mfma'): d['mfma'] += 1
    elif op.startswith('ds_read'): d['ds_rd'] += 1
    elif op.startswith('ds_write'): d['ds_wr'] += 1
    elif op.startswith('buffer_load') or op.startswith('global_load'): d['vm_ld'] += 1
    elif op.startswith('buffer_store') or op.startswith('global_store'): d['vm_st'] += 1
    elif op.startswith('scratch'): d['scratch'] += 1
    elif op.startswith('v_accvgpr'): d['acc_mov'] += 1
    elif op.startswith('v_pk'): d['v_pk'] += 1
    elif op.startswith('v_add_u32') or op.startswith('v_add3') or op.startswith('v_lshl_add') or op.startswith('v_add_lshl') or op.startswith('v_or_b32') or op.startswith('v_lshl_or'): d['v_int'] += 1
    elif op.startswith('v_'): d['v_other'] += 1
    elif op.startswith('s_waitcnt'): d['wait'] += 1
    elif op.startswith('s_barrier'): d['barrier'] += 1
    elif op.startswith('s_nop'): d['nop'] += 1
    elif op.startswith('s_'): d['salu'] += 1
keys = ['mfma', 'v_pk', 'v_int', 'v_other', 'acc_mov', 'ds_rd', 'ds_wr', 'vm_ld', 'vm_st', 'scratch', 'wait', 'barrier', 'nop', 'salu']
print('%-12s' % 'block' + ''.join('%8s' % k for k in keys))
for b, d in stats.items():
    if sum(d.values()) >= floor:
        print('%-12s' % b + ''.join('%8d' % d.get(k, 0) for k in keys))
