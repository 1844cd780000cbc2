"""Developer tool: per-phase shader cycles of the MFMA backward (library built with
LP_BUILD_FLAGS=-DLP_PHASE_TIMING, selected with LIGHTPLANE_AMD_LIB).  Prints average cycles per (wave, sample) for every
phase of the kernel the launcher picks (LP_BF3_BWD=1: the bf16x3 backward)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from bench import RendererWorkload

dev = torch.device("cuda:0")
lp.config.check_inputs = False
wl = RendererWorkload(os.environ.get("LP_PHASE_WORKLOAD", "cfg2"), 0, dev, None, _lib.LP_KERNEL_AUTO)
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.lp_debug_phase_cycles.argtypes = [ctypes.c_void_p]
n = 5
for it in range(n + 1):
    if it == 1:
        assert L.lp_debug_phase_cycles(buf) == 0, "library not built with -DLP_PHASE_TIMING"
    wl.step()
rc = L.lp_debug_phase_cycles(buf)
print('rc', rc, L.lp_last_error(), {k: v for k, v in os.environ.items() if k.startswith("LP_")})
names = ["fwd", "compositing", "heads_bwd", "c1", "o1", "t2", "t1", "fetch", "scatter", "outside"]
waves = (wl.n_rays + 31) // 32
tot = 0
for i, nm in enumerate(names):
    c = buf[i] / (n * waves * wl.S)
    tot += c
    print(f"{nm:12s} {c:10.0f} cycles / wave-sample")
print(f"{'total':12s} {tot:10.0f}")
