#!/usr/bin/env python3
# instruction mix of the large basic blocks of one kernel in an assembly listing: bbstat.py file.s kernel-name-regex [min]
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^(' + sys.argv[2] + r'\w*):', s, re.M)
a = m.start(); b = s.index('.Lfunc_end', a)
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
blocks = []; cur = None
for l in s[a:b].splitlines():
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = [l.strip(), 0, collections.Counter()]; blocks.append(cur)
    elif cur and l.startswith('\t') and not l.strip().startswith(('.', ';')):
        cur[1] += 1; cur[2][l.split()[0]] += 1
for bl in blocks:
    if bl[1] > mn: print(bl[0], bl[1], bl[2].most_common(30))
