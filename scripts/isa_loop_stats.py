"""Instruction mix of the loops of one kernel in a hipcc -S listing (design aid)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print('kernel lines', len(body), 'loops', loops)
for a, b in loops:
    ins = [l.split()[0] for l in body[a:b + 1]
           if l.strip() and not l.strip().startswith(('.', ';')) and not re.match(r'^\.LBB', l)]
    c = collections.Counter(ins)
    f64 = sum(v for k, v in c.items() if 'f64' in k)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(f'loop {a}-{b}: {len(ins)} instr, {valu} VALU, {f64} f64')
    print('   ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(30)))
