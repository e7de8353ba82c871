#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in a hipcc -S listing.
usage: isa_blocks.py <file.s> <mangled-name-prefix> [max-blocks]"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
pre = sys.argv[2]
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 8
start = [i for i, l in enumerate(lines) if l.startswith(pre) and ':' in l][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
seg = []; cur = ('entry', [])
for l in lines[start + 1:end]:
    s = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', s):
        seg.append(cur); cur = (s.split(':')[0], [])
    elif s and not s.startswith(';'):
        cur[1].append(s)
seg.append(cur)
keys = ('accvgpr', 'mfma', 'scratch', 'lds', 'barrier', 'branch', 'nop', 'waitcnt', 'ds_read', 'global_')
for name, ins in seg[:nmax]:
    c = collections.Counter(x.split()[0] for x in ins)
    print(name, len(ins), {k: v for k, v in c.items() if any(t in k for t in keys)})
for l in lines[end:end + 60]:
    if 'NumVgprs' in l or 'NumAgprs' in l or 'ScratchSize' in l or 'NumSgprs' in l:
        print(l)
