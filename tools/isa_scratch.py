#!/usr/bin/env python3
"""Where a kernel's scratch (spill) traffic sits: every scratch_load / scratch_store of one kernel in a `hipcc -S` dump with
the innermost loop (back-branch range) that contains it, and that loop's length.  Spills outside every loop run once per
work item; spills inside a solver loop are the ones that cost.  Usage: tools/isa_scratch.py <file.s> <kernel name pattern>"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
ninstr = lambda a, b: sum(1 for x in body[a:b] if x.startswith('\t') and not x.startswith('\t.'))
by = collections.Counter()
for i, l in enumerate(body):
    if 'scratch_' in l:
        inner = min(((b - a, a, b, n) for a, b, n in loops if a < i <= b), default=None)
        key = ('outside loops', 0) if inner is None else (f'{inner[3]} (len {ninstr(inner[1], inner[2])})', inner[1])
        by[(key, 'load' if 'scratch_load' in l else 'store')] += 1
print(lines[start].split(':')[0], 'instructions', ninstr(0, len(body)), 'scratch ops', sum(by.values()))
for (key, kind), n in sorted(by.items(), key=lambda kv: kv[0][0][1]):
    print(f'  {key[0]:36s} {kind:5s} {n}')

if '--defs' in sys.argv:   # every spill store with the instruction that produced the stored register
    for i, l in enumerate(body):
        if 'scratch_store' in l:
            reg = re.search(r'scratch_store_dword(?:x\d)?\s+off,\s*(v\[?[\d:]+\]?)', l).group(1)
            base = re.match(r'v\[?(\d+)', reg).group(1)
            d = next((body[j].strip() for j in range(i - 1, max(0, i - 600), -1)
                      if re.search(r'^\s+\S+\s+v\[?' + base + r'\b', body[j]) and 'scratch' not in body[j]), '?')
            print(f'  {i:6d} {l.strip()[:64]:64s} <= {d[:80]}')
