#!/usr/bin/env python3
"""List, for one kernel symbol of a device assembly file (hipcc -S --cuda-device-only x.hip -o x.s), every global / flat / buffer load whose
value is waited for with s_waitcnt vmcnt(0) before another load is issued: dependent round trips the compiler may have created by sinking a
load behind a test or by turning a select of two values into a load from a selected address.  usage: isa_scan.py x.s <mangled symbol prefix>"""
import sys, re
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^' + pat + r'.*:\s*;?.*$', l) and not l.startswith('\t'):
        start = i; break
if start is None: print('no symbol', pat); sys.exit()
end = start
while 's_endpgm' not in lines[end]: end += 1
body = [l for l in lines[start:end+1] if l.startswith('\t') and not l.strip().startswith(';')]
# immediate waits: load followed by vmcnt(0) within 12 instrs with no other load in between
out = []
for i, l in enumerate(body):
    if re.search(r'\b(global|flat|buffer)_load', l):
        for j in range(i+1, min(i+14, len(body))):
            if re.search(r'\b(global|flat|buffer)_load', body[j]): break
            if 's_waitcnt' in body[j] and 'vmcnt(0)' in body[j]:
                out.append((i, l.strip()[:70], j - i)); break
print(pat, 'instructions', len(body), 'immediate-wait loads:', len(out))
for i, l, d in out[:40]: print('  @%d +%d  %s' % (i, d, l))
