#!/usr/bin/env python
"""Per-section instruction statistics of one kernel: compiles a .hip with -DLP_ASM_MARKS (asm comment
markers `; LPMARK name`) and counts opcode classes between consecutive markers.
usage: isa_sections.py file.hip 'kernel-name-substring' [extra flags]"""
import os, re, subprocess, sys, collections
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
from lightplane_amd.csrc import build as B
src, kname = sys.argv[1], sys.argv[2]
out = "/tmp/isa_sections.s"
cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DLP_ASM_MARKS", "-S", "--cuda-device-only", os.path.join(B.HERE, src), "-o", out] + sys.argv[3:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and kname in subprocess.run(["c++filt", l.split(":")[0]], stdout=subprocess.PIPE).stdout.decode())
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
classes = [("mfma", r"v_mfma"), ("scratch", r"scratch_"), ("ds_read", r"ds_read|ds_bpermute"), ("ds_write", r"ds_write|ds_add"), ("vmem", r"global_load|buffer_load"),
           ("atomic", r"global_atomic"), ("gstore", r"global_store"), ("waitcnt", r"s_waitcnt"), ("branch", r"s_cbranch|s_branch"), ("valu", r"v_"), ("salu", r"s_")]
sec, stats, order = "prologue", collections.defaultdict(collections.Counter), ["prologue"]
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; LPMARK (\S+)", t)
    if m:
        sec = m.group(1)
        if sec not in order: order.append(sec)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    op = t.split()[0]
    for c, pat in classes:
        if re.match(pat, op):
            stats[sec][c] += 1
            break
    else:
        stats[sec]["other"] += 1
cols = [c for c, _ in classes] + ["other"]
print(f"{'section':14s}" + "".join(f"{c:>9s}" for c in cols))
for s_ in order:
    print(f"{s_:14s}" + "".join(f"{stats[s_][c]:9d}" for c in cols))
tot = collections.Counter()
for s_ in order: tot.update(stats[s_])
print(f"{'TOTAL':14s}" + "".join(f"{tot[c]:9d}" for c in cols))
