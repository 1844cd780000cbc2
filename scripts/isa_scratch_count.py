#!/usr/bin/env python
"""Scratch INSTRUCTIONS per kernel of one .hip source (compiled to ISA with the build's flags for that file): the compiler's
resource remark reports the scratch FRAME, which can be non-zero for a kernel that never touches it (a stack object that was
optimised away after the frame was sized) -- e.g. the deep layer-looped backward's triplane instantiations: 164 B frame, 0
scratch instructions.    python scripts/isa_scratch_count.py lp_renderer_loop.hip [kernel-name-substring]"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
from lightplane_amd.csrc import build as B
src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/isa_scratch_count.s"
cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.FILE_FLAGS.get(src, []) + ["-S", "--cuda-device-only", os.path.join(B.HERE, src), "-o", out]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
frames = {}
cur = None
for l in lines:
    m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
    if m: cur = m.group(1)
    m = re.match(r"\s*\.amdhsa_private_segment_fixed_size (\d+)", l)
    if m and cur: frames[cur] = int(m.group(1))
print(f"# {' '.join(cmd[:-2])} ...")
print(f"{'kernel':72s} {'frame B':>8s} {'scratch insts':>14s}")
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\S+):\s", l + " ")
    if not m or m.group(1) not in frames:
        continue
    sym = m.group(1)
    name = re.sub(r"\(.*", "", subprocess.run(["c++filt", sym], stdout=subprocess.PIPE).stdout.decode().strip()).replace("void lp::", "")
    if sub not in name:
        continue
    end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
    n = sum(1 for x in lines[i:end] if "scratch_" in x)
    print(f"{name:72s} {frames[sym]:8d} {n:14d}")
