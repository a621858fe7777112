#!/usr/bin/env python3
"""Summarise the loops of one kernel in a hipcc -S dump: instruction mix per loop body."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
print(lines[start].split(':')[0], 'instructions', sum(1 for l in body if l.startswith('\t') and not l.startswith('\t.')))
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        b = body[labels[m.group(1)]:i]
        c = lambda p: sum(1 for x in b if re.search(p, x))
        print('  loop %-10s len %5d | fma64 %4d mul64 %3d add64 %3d | ds_read %3d ds_write %3d bperm %2d | scratch %3d | barrier %d | rcp/div %2d | waitcnt %3d | readlane %d' % (
            m.group(1), len(b), c('v_fma_f64|v_fmac_f64'), c('v_mul_f64'), c('v_add_f64'), c('ds_read'), c('ds_write'), c('ds_bpermute|ds_swizzle'), c('scratch_'), c('s_barrier'), c('v_rcp_f64|v_div_'), c('s_waitcnt'), c('v_readlane|v_readfirstlane')))
