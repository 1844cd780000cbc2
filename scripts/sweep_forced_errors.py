"""Forced-oracle proof over EVERY Renderer sweep case of tests/test_gpu_sweep.py (random sweep, segmented sweep, the reference's own
axes): per case the kernel's ReLU decisions (DUMP twin) are forced onto the fp64 oracle on the reference's fp32 geometry; prints the
worst entry over outputs + every gradient, the number of forced units and their largest margin.  The data behind the sweep's bars.
    python scripts/sweep_forced_errors.py > gpurun_out/sweep_forced.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
warnings.filterwarnings("ignore")
import torch
import lightplane_amd as lp
lp.config.warn_generic_kernel = False
from tests.test_gpu_sweep import _renderer_case, _segmented_case, _reference_axes_case
from tests import test_gpu_parity as P
dev = torch.device("cuda:0")
rows = []
for fam, make, n in (("sweep", _renderer_case, 48), ("segsweep", _segmented_case, 16), ("refsweep", _reference_axes_case, 40)):
    for i in range(n):
        case = make(i)
        d = case.build()
        if not P.has_dump_twin(d):
            rows.append(dict(case=case.name, twin=False, family=lp.kernel_family(d["rays"], d["grids"], d["decoder"], color_grid=d["color_grids"],
                                                                                  num_samples_inf=case.num_samples_inf)))
            continue
        del P.FORCED_EVENTS[:]
        err = None
        try:
            P.forced_oracle_check(case.name, d, dev, chunk=d["rays"].n_rays, tol=1e-4)
        except AssertionError as e:
            err = str(e)[:300]
        ev = P.FORCED_EVENTS[-1] if P.FORCED_EVENTS else {}
        rows.append(dict(case=case.name, twin=True, inf=case.num_samples_inf, forced=ev.get("forced_units"), margin=ev.get("max_forced_margin"),
                         worst=max(ev.get("worst", {"-": None}).values()) if ev.get("worst") else None,
                         worst_tensor=max(ev["worst"], key=ev["worst"].get) if ev.get("worst") else None, error=err))
ok = [r for r in rows if r.get("twin") and not r.get("error")]
print(json.dumps(dict(n_cases=len(rows), n_with_twin=sum(bool(r.get("twin")) for r in rows), n_proven_at_1e4=len(ok),
                      failures=[r for r in rows if r.get("error")], no_twin=[r for r in rows if not r.get("twin")],
                      worst_entry=max((r["worst"] for r in ok), default=None), largest_margin=max((r["margin"] for r in ok), default=None),
                      rows=rows), indent=1))
