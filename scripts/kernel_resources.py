#!/usr/bin/env python
"""Print a table of per-kernel register / scratch / occupancy figures of one .hip source
(hipcc -Rpass-analysis=kernel-resource-usage), with the build's own flags (the per-file ones of build.py FILE_FLAGS included)."""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from lightplane_amd.csrc import build as B

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.FILE_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.HERE, src), "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+:\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
    else:
        kv = m.group(1).split(": ")
        if len(kv) != 2:
            continue
        k, v = kv
    if k.endswith("Name"):
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE).stdout.decode().strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k.strip()] = v
print(f"{'kernel':70s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'vspill':>7s}")
for r in rows:
    n = re.sub(r"\(.*", "", r["name"]).replace("void lp::", "")
    print(f"{n:70s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('VGPRs Spill','?'):>7s}")
