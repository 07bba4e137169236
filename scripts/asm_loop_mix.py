"""Instruction mix of a kernel's main loop = the backward-branch region that holds the most MFMAs:
python scripts/asm_loop_mix.py FILE.s MANGLED_NAME"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
labels = {l.split(':')[0].strip(): i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo = labels[m.group(1)]
        n = sum('v_mfma' in x for x in body[lo:i])
        if best is None or n > best[0] or (n == best[0] and i - lo < best[2] - best[1]):
            best = (n, lo, i)
_, start, end = best
ops = {}
for l in body[start:end]:
    t = l.strip().split()
    if not t or t[0].startswith(';') or t[0].endswith(':') or t[0].startswith('.'):
        continue
    ops[t[0]] = ops.get(t[0], 0) + 1
print(name[:40], 'lines', start, end, 'instructions', sum(ops.values()))
print('  ' + '  '.join('%s %d' % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])[:16]))
