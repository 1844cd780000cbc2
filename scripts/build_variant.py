#!/usr/bin/env python
"""A/B library builds: ab/lib<name>.so = the product objects (lightplane_amd/csrc/build/*.o, built first) with SOME translation
units recompiled with extra flags.    python scripts/build_variant.py NAME file.hip:"-DX=1 -DY=2" [file2.hip:"..."]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
from lightplane_amd.csrc import build as B
B.build()
name = sys.argv[1]
over = dict(a.split(":", 1) for a in sys.argv[2:])
ab = os.path.join(os.path.dirname(HERE), "ab"); os.makedirs(os.path.join(ab, "obj_" + name), exist_ok=True)
objs, procs = [], []
for src in B.SOURCES:
    o = os.path.join(B.HERE, "build", src.replace(".hip", ".o"))
    if src in over:
        o = os.path.join(ab, "obj_" + name, src.replace(".hip", ".o"))
        cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.FILE_FLAGS.get(src, []) + over[src].split() + ["-c", os.path.join(B.HERE, src), "-o", o]
        procs.append(subprocess.Popen(cmd))
    objs.append(o)
assert all(p.wait() == 0 for p in procs)
out = os.path.join(ab, f"lib{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
