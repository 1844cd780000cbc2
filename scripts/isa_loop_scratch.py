#!/usr/bin/env python
"""Where are the scratch (spill) instructions of the dominant Renderer backward kernel?  Compiles lp_renderer_mfma_bwd.hip with
the build's own flags (-DLP_DEV_ONE: the one instantiation the headline runs, renderer_bwd_bf3<16, 1, true, 3, 4, false>), finds
the sample loop (the largest backward branch) and lists the scratch instructions inside / outside it.  No GPU needed.
    python scripts/isa_loop_scratch.py > profiles/r03_bwd_scratch_in_loop.txt"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
from lightplane_amd.csrc import build as B
src = "lp_renderer_mfma_bwd.hip"
out = "/tmp/isa_loop_scratch.s"
cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.FILE_FLAGS.get(src, []) + ["-DLP_DEV_ONE", "-S", "--cuda-device-only", os.path.join(B.HERE, src), "-o", out]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2lp16renderer_bwd_bf3ILi16ELi1ELb1ELi3ELi4ELb0E") and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
scr = [i for i, l in enumerate(body) if "scratch_" in l]
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
a, b = max(loops, key=lambda x: x[1] - x[0])
frame = next(l.split()[-1] for l in lines[end:] if "private_segment_fixed_size" in l)
print("command:", " ".join(cmd[:-2] + ["..."]))
print(f"kernel renderer_bwd_bf3<16, 1, true, 3, 4, false>: {len(body)} lines of ISA, scratch frame {frame} B per lane, "
      f"{len(scr)} scratch instructions in the whole kernel")
print(f"sample loop = lines {a}..{b} of the kernel ({b - a} lines)")
inside = [i for i in scr if a <= i <= b]
print(f"scratch instructions INSIDE the sample loop: {len(inside)}")
for i in inside:
    print("   ", body[i].strip())
print(f"scratch instructions outside (prologue / epilogue): {len(scr) - len(inside)}")
for i in scr:
    if i not in inside:
        print("   ", body[i].strip())
