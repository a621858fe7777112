#!/usr/bin/env python3
"""Which kernels differ between two `hipcc -S --cuda-device-only` dumps of rg_mpc.hip (labels normalised): after a change that
is meant to touch one kernel, the proof that every other kernel is the same code.  Usage: tools/isa_identity.py <a.s> <b.s>"""
import re, sys
def kernels(path):
    lines = open(path).read().split('\n')
    out, name, body = {}, None, []
    for l in lines:
        m = re.match(r'^(_Z\S+|rg_\w+):\s*(;.*)?$', l)
        if m and name is None and not l.startswith('.'):
            name, body = m.group(1), []
            continue
        if name is not None:
            if l.startswith('.Lfunc_end'):
                out[name] = body; name = None
            elif l.startswith('\t') and not l.startswith('\t.') and not l.strip().startswith(';'):
                body.append(re.sub(r'\.Lpost_getpc\d+', '.Lpost_getpc', re.sub(r'\.LBB\d+_', '.LBB_', l.split(';')[0].rstrip())))
    return out
a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
for k in sorted(set(a) | set(b)):
    if k not in a or k not in b: print('ONLY IN ONE', k)
    else: print(('identical ' if a[k] == b[k] else 'DIFFERENT ') + f'{len(a[k]):6d} {len(b[k]):6d}  {k[:110]}')
