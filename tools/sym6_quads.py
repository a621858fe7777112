#!/usr/bin/env python3
"""Lane numbering for the quad-spread publish of the symmetric sweep study (tests/studies/sweep_variants.inc: Sym6Quads): the
NB (NB + 1) / 2 blocks (r, c), r >= c, of the lower block triangle cut into groups of at most four (the lanes of a DPP quad) such
that no two blocks of a group share an index -- then at most one lane of a quad publishes in any pivot step and the other lanes
can store its second and third pair.  Randomised greedy; prints the packed tables (one byte per lane: br << 4 | bc, 0xFF = no
block; eight lanes per 64-bit word) and the lanes of the diagonal blocks, six bits each.  Usage: tools/sym6_quads.py [NB=10]"""
import random
import sys

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 10
LANES = 64 if NB <= 10 else 256
assert NB <= 15, "one nibble per block index"
blocks = [(r, c) for r in range(NB) for c in range(r + 1)]
random.seed(1)
for trial in range(100000):
    random.shuffle(blocks)
    quads = [[] for _ in range(LANES // 4)]
    for blk in blocks:
        cands = [q for q in quads if len(q) < 4 and all(not (set(blk) & set(o)) for o in q)]
        if not cands:
            break
        cands.sort(key=lambda q: -len(q))
        random.choice(cands[:3]).append(blk)
    else:
        break
else:
    sys.exit("no assignment found")
lanes = [b for q in quads for b in (q + [None] * (4 - len(q)))]
for kb in range(NB):   # at most one publisher per quad and step
    assert all(sum(1 for b in q if kb in b) <= 1 for q in quads)
codes = [0xFF if b is None else (b[0] << 4 | b[1]) for b in lanes]
words = [sum(codes[8 * w + i] << (8 * i) for i in range(8)) for w in range(LANES // 8)]
print("MAP  = {" + ", ".join("0x%016XULL" % w for w in words) + "}")
print("DIAG = 0x%016XULL   // lanes %s" % (sum(lanes.index((k, k)) << (6 * k) for k in range(NB)), [lanes.index((k, k)) for k in range(NB)]))
