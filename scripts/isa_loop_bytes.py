"""Instruction mix AND encoding sizes of the hot loop of one kernel, from the disassembly of the
code object (design aid).  A lone wave issues a 4-byte instruction every ~1.85 ns and an 8-byte one
every ~2.24 ns (scripts/ubench/f64_forms.hip), so the bytes are the cost.

  hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 csrc/swimmer_kernels.hip -o /tmp/k.hsaco
  clang-offload-bundler --unbundle --type=o --input=/tmp/k.hsaco \
        --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/k.co
  llvm-objdump -d /tmp/k.co > /tmp/k.dis
  python scripts/isa_loop_bytes.py /tmp/k.dis rollout_quad3_kernelILb1ELb1ELb1E [steps_per_trip]
"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <\S*" + pat + r"\S*>:", l))
body = []
for l in lines[start + 1:]:
    if re.match(r"^[0-9a-f]+ <", l):
        break
    m = re.match(r"^\s+(\S+)(.*?)//\s*([0-9A-F]+):\s*((?:[0-9A-F]{8}\s*)+)", l)
    if m:
        body.append((int(m.group(3), 16), m.group(1), len(m.group(4).split()) * 4, l))
addr_index = {a: i for i, (a, *_rest) in enumerate(body)}
# backward branches: target address < own address
loops = []
for i, (a, op, size, text) in enumerate(body):
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"<\S+\+0x([0-9a-f]+)>", text)
        if m:
            tgt = int(m.group(1), 16) + body[0][0]
            if tgt in addr_index and tgt < a:
                loops.append((addr_index[tgt], i))
if not loops:
    sys.exit("no loops")


def score(ab):
    ins = body[ab[0]:ab[1] + 1]
    return (sum("dpp" in t[3] for t in ins), sum("f64" in t[1] for t in ins))


innermost = [ab for ab in loops if not any(o != ab and ab[0] <= o[0] and o[1] <= ab[1] for o in loops)]
a, b = max(innermost, key=score)
ins = body[a:b + 1]
# instructions behind a forward branch that jumps OUT of the loop body are never in the stream;
# the out-of-line re-normalisation blocks live outside [a, b] already
by_size = collections.Counter()
by_op = collections.Counter()
for _, op, size, _ in ins:
    by_size[size] += 1
    by_op[(op, size)] += 1
n = len(ins)
tot = sum(s * c for s, c in by_size.items())
print(f"{pat}: hot loop {n} instructions, {tot} bytes per trip "
      f"({n / steps:.1f} instructions, {tot / steps:.0f} bytes per step at {steps} steps per trip)")
print("   by encoding size: " + ", ".join(f"{s} B: {c} ({c / steps:.1f}/step)" for s, c in sorted(by_size.items())))
print("   " + ", ".join(f"{op}[{s}B]:{c}" for (op, s), c in by_op.most_common(24)))
