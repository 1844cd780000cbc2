#!/usr/bin/env python
"""Build liblightplane_hip.so for gfx950 with hipcc (no cmake, no torch extension machinery).

    python lightplane_amd/csrc/build.py [--force] [--verbose]

Output: lightplane_amd/liblightplane_hip.so (git-ignored; travels to the GPU box with gpurun).
Flags: -ffp-contract=off keeps the coordinate/index arithmetic individually rounded (bit-exact
integer indexing vs the oracle); FMAs in the MLP / interpolation math are explicit fmaf().
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "liblightplane_hip.so")
SOURCES = ["lp_api.hip", "lp_renderer_generic.hip", "lp_renderer_mfma.hip", "lp_renderer_mfma_bwd.hip", "lp_renderer_loop.hip", "lp_renderer_loop_shallow.hip", "lp_splatter.hip", "lp_splatter_mlp.hip", "lp_splatter_mlp_loop.hip", "lp_splatter_mlp_loop_shallow.hip", "lp_ray_embedding.hip"]
HEADERS = ["lp_device.h", "lp_host.h", "lp_mfma_common.h", "lp_generic_mlp.h", "lp_splat_walk.h", "lp_bf3.h", "lp_loop.h", "lp_renderer_loop.h", "lp_splatter_mlp_loop.h", os.path.join("..", "..", "include", "lightplane_hip.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
]


FLAGS += os.environ.get("LP_BUILD_FLAGS", "").split()

# Per-file flags.  The Renderer backward sits at the register limit (256 VGPRs at two waves per SIMD); MachineLICM hoists
# ~25 loop-invariant address / constant computations out of the sample loop, which the allocator then spills and
# reloads inside it -- and a reload costs ~300 cycles there.  Without MachineLICM: 28 instead of 52
# spilled registers, 22 instead of 34 scratch instructions per sample; with the register allocator additionally allowed
# to sink (= recompute at the use) instead of spilling: 4 spilled registers and NO scratch instruction in the sample loop,
# for ~280 rematerialised VALU instructions (of ~3 400 per sample).
FILE_FLAGS = {
    "lp_renderer_mfma_bwd.hip": os.environ.get("LP_BWD_FLAGS", "-mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills=1").split(),
    # the shallow two-waves-per-SIMD backward of the layer-looped family: the same two switches take it from 49 spilled
    # registers to none (they cost the deep one-wave instantiations of lp_renderer_loop.hip 1-2 %, so those keep the defaults)
    # (-DLP_LOOP_DW_FP32: the two-waves-per-SIMD instantiations keep the fp32 weight-gradient quadrants -- with the bf16 ones of
    # lp_loop.h the 32-channel shallow backwards spill 16-27 registers; the one-wave instantiations of lp_renderer_loop.hip /
    # lp_splatter_mlp_loop.hip have the room)
    "lp_renderer_loop_shallow.hip": os.environ.get("LP_LOOP_FLAGS", "-mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills=1").split() + ["-DLP_LOOP_DW_FP32"],
    "lp_splatter_mlp_loop_shallow.hip": os.environ.get("LP_LOOP_FLAGS", "-mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills=1").split() + ["-DLP_LOOP_DW_FP32"],
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed ---\n{out}\n")
        elif out.strip() and verbose:
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
