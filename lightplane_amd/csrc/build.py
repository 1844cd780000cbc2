#!/usr/bin/env python
"""Build liblightplane_hip.so for gfx950 with hipcc (no cmake, no torch extension machinery).

    python lightplane_amd/csrc/build.py [--force] [--verbose]

Output: lightplane_amd/liblightplane_hip.so (git-ignored; travels to the GPU box with gpurun).
Flags: -ffp-contract=off keeps the coordinate/index arithmetic individually rounded (bit-exact
integer indexing vs the oracle); FMAs in the MLP / interpolation math are explicit fmaf().
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "liblightplane_hip.so")
# (longest compiles first: the translation units are compiled in parallel)
SOURCES = ["lp_renderer_loop.hip", "lp_renderer_loop_dump.hip", "lp_renderer_mfma_bwd.hip", "lp_renderer_mfma_bwd_dump.hip", "lp_renderer_mfma_bwd_c32.hip", "lp_renderer_mfma_bwd_aux.hip", "lp_renderer_mfma_bwd_tm.hip",
           "lp_splatter_mlp_loop.hip", "lp_renderer_mfma.hip", "lp_renderer_loop_shallow.hip", "lp_renderer_loop_shallow_dump.hip",
           "lp_renderer_generic.hip", "lp_splatter.hip", "lp_splatter_mlp.hip", "lp_splatter_mlp_loop_shallow.hip", "lp_ray_embedding.hip", "lp_api.hip"]
HEADERS = ["lp_device.h", "lp_host.h", "lp_mfma_common.h", "lp_generic_mlp.h", "lp_splat_walk.h", "lp_bf3.h", "lp_loop.h", "lp_renderer_loop.h",
           "lp_renderer_mfma_bwd.h", "lp_splatter_mlp_loop.h", os.path.join("..", "..", "include", "lightplane_hip.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
]


# -DLP_TEST_HOOKS: the DUMP twins behind lp_renderer_backward_relu_dump() (their own translation units; the production kernels are the
# same with or without them).  LP_NO_TEST_HOOKS=1 builds a library without them (the hook then returns LP_EUNSUPPORTED).
if not os.environ.get("LP_NO_TEST_HOOKS"):
    FLAGS.append("-DLP_TEST_HOOKS")
FLAGS += os.environ.get("LP_BUILD_FLAGS", "").split()

# Per-file flags.  The Renderer backward sits at the register limit (256 VGPRs at two waves per SIMD); MachineLICM hoists
# ~25 loop-invariant address / constant computations out of the sample loop, which the allocator then spills and
# reloads inside it -- and a reload costs ~300 cycles there.  Without MachineLICM: 28 instead of 52
# spilled registers, 22 instead of 34 scratch instructions per sample; with the register allocator additionally allowed
# to sink (= recompute at the use) instead of spilling: 4 spilled registers and NO scratch instruction in the sample loop,
# for ~280 rematerialised VALU instructions (of ~3 400 per sample).
_BWD_FLAGS = os.environ.get("LP_BWD_FLAGS", "-mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills=1").split()
_LOOP_FLAGS = os.environ.get("LP_LOOP_FLAGS", "-mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills=1").split() + ["-DLP_LOOP_DW_FP32"]
FILE_FLAGS = {
    # (the four translation units of the tuned backward -- 16 channels, 32 channels, LP_ARITH_FP32, DUMP twins -- share their flags: a
    # DUMP twin is its production kernel + stores)
    "lp_renderer_mfma_bwd.hip": _BWD_FLAGS,
    "lp_renderer_mfma_bwd_c32.hip": _BWD_FLAGS,
    "lp_renderer_mfma_bwd_aux.hip": _BWD_FLAGS,
    "lp_renderer_mfma_bwd_dump.hip": _BWD_FLAGS,
    "lp_renderer_mfma_bwd_tm.hip": _BWD_FLAGS,
    # the shallow two-waves-per-SIMD backward of the layer-looped family: the same two switches take it from 49 spilled
    # registers to none (they cost the deep one-wave instantiations of lp_renderer_loop.hip 1-2 %, so those keep the defaults)
    # (-DLP_LOOP_DW_FP32: the two-waves-per-SIMD instantiations keep the fp32 weight-gradient quadrants -- with the bf16 ones of
    # lp_loop.h the 32-channel shallow backwards spill 16-27 registers; the one-wave instantiations of lp_renderer_loop.hip /
    # lp_splatter_mlp_loop.hip have the room)
    "lp_renderer_loop_shallow.hip": _LOOP_FLAGS,
    "lp_renderer_loop_shallow_dump.hip": _LOOP_FLAGS,
    "lp_splatter_mlp_loop_shallow.hip": _LOOP_FLAGS,
}


def source_hash():
    """sha256 over everything the library is compiled from: csrc/*.hip, csrc/*.h, this script and include/lightplane_hip.h (names and
    contents, sorted).  lp_build_info() carries the value the binary was built from; __graft_entry__.build() and bench.py compare it
    with the tree."""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(HERE) if f.endswith((".hip", ".h")) or f == "build.py")
    paths = [os.path.join(HERE, f) for f in files] + [os.path.join(HERE, "..", "..", "include", "lightplane_hip.h")]
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _write_gen_header(objdir):
    """build/lp_build_gen.h for lp_api.hip (lp_build_info): rewritten only when its content changes, so lp_api.o is rebuilt exactly then."""
    flags = {"all": FLAGS, "per_file": {k: v for k, v in FILE_FLAGS.items() if v}}
    esc = json.dumps(json.dumps(flags))  # a C string literal holding the JSON text
    text = f'#define LP_BUILD_SRC_HASH "{source_hash()}"\n#define LP_BUILD_FLAGS_JSON {esc}\n'
    path = os.path.join(objdir, "lp_build_gen.h")
    old = open(path).read() if os.path.exists(path) else None
    if old != text:
        with open(path, "w") as f:
            f.write(text)
    return path


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
    gen = _write_gen_header(objdir)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs + ([gen] if src == "lp_api.hip" else [])):
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
