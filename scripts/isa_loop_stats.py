"""Instruction mix of the hot loop of one kernel in a hipcc -S listing (design aid): the per-step
instruction counts quoted in DESIGN.md.

  hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only -I../../include \
        swimmer_kernels.hip -o /tmp/k.s
  python scripts/isa_loop_stats.py /tmp/k.s 'rollout_quad3_kernelILb1ELb1ELb1E'

The hot loop is the innermost loop (no other backward branch inside) with the most DPP instructions (the rollout
kernels) or, failing that, the most f64 instructions."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = [l.split(';')[0].rstrip() for l in lines[start:end + 1] if l.split(';')[0].strip()]
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
if not loops:
    sys.exit(f'{pat}: no loops in {len(body)} lines')


def instrs(a, b):
    return [l.split()[0] for l in body[a:b + 1] if not l.startswith('.')]


def score(ab):
    ins = instrs(*ab)
    dpp = sum('dpp' in x for x in ins)
    f64 = sum('f64' in x for x in ins)
    # innermost: prefer the shorter of two loops with the same content
    return (dpp, f64, -len(ins))


innermost = [ab for ab in loops
             if not any(o != ab and ab[0] <= o[0] and o[1] <= ab[1] for o in loops)]
a, b = max(innermost, key=score)
ins = instrs(a, b)
c = collections.Counter(ins)
f64 = sum(v for k, v in c.items() if 'f64' in k)
valu = sum(v for k, v in c.items() if k.startswith('v_'))
print(f'{pat}: {len(body)} lines, {len(loops)} backward branches; hot loop lines {a}-{b}: '
      f'{len(ins)} instructions, {valu} VALU, {f64} f64')
print('   ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(30)))
