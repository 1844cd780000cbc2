#!/usr/bin/env python
"""List the vector-memory instructions (loads, scratch, atomics) and the vmcnt waits of one kernel in program
order, with instruction indices: shows at a glance whether loads are batched or serialised and where spill
reloads drain outstanding atomics.  usage: isa_memops.py file.s mangled-kernel-prefix [first [last]]"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
pref = sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
k = 0
for i in range(start, end):
    t = lines[i].strip()
    if not t or (t.startswith((";", ".")) and not t.startswith(".LBB")):
        continue
    k += 1
    if lo <= k <= hi and re.match(r"scratch_|s_waitcnt.*vmcnt|global_load|buffer_load|global_atomic|s_barrier", t):
        print(k, t[:72])
