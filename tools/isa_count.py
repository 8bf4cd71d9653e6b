"""Static instruction mix per kernel of a device-only assembly file (hipcc -S --cuda-device-only): python3 tools/isa_count.py file.s [filter]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'^(_Z\S+):\s*; @\S+\n', s, flags=re.M)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('s_endpgm')[0]
    if flt not in name:
        continue
    c = Counter()
    prev = ""
    for l in body.split('\n'):
        if not l.startswith('\t'):
            continue
        t = l.strip().split()
        if not t or t[0].startswith(('.', ';')):
            continue
        op = t[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
            if prev.startswith('s_nop'):
                c['nop_before_mfma'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('s_nop'):
            c['s_nop'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            c['vmem'] += 1
        prev = op
    print(name[-64:], dict(c))
