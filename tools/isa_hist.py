"""Instruction-class histogram per basic block of one kernel in a device assembly file (hipcc --cuda-device-only -S):
    python tools/isa_hist.py file.s kernel_name_substring [min_block_len]"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*%s\S*:' % re.escape(key), l))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
blocks, cur, name = [], [], 'entry'
for l in s[start + 1:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.strip(); cur = []
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur.append(t.split()[0])
blocks.append((name, cur))
def cls(i):
    if i.startswith('v_mfma'): return 'mfma'
    if i.startswith('v_'): return 'valu'
    if i.startswith('ds_'): return 'lds'
    if i.startswith('global_') or i.startswith('buffer_') or i.startswith('scratch_'): return 'vmem'
    if i.startswith('s_waitcnt'): return 'waitcnt'
    if i.startswith('s_barrier'): return 'barrier'
    if i.startswith('s_nop'): return 'nop'
    if i.startswith('s_'): return 'salu'
    return 'other'
print(s[start][:100])
for n, c in blocks:
    if len(c) < minlen: continue
    print(n, len(c), dict(collections.Counter(cls(i) for i in c)))
    v = collections.Counter(i for i in c if cls(i) == 'valu')
    print('    valu:', v.most_common(12))
