"""Does the ORDER of the rays of an image matter?  The kernels merge consecutive rays that fall into the same grid cell (scatter
walks) and a wave gathers for 32 consecutive rays: row-major pixels put a 32 x 1 strip of the image into a wave, tile order
(lightplane_amd.rays.tile_order) a th x tw block.  cfg-2 workload (Renderer) and cfg-3 workload (Splatter), forward + backward."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from bench import RendererWorkload, SplatterWorkload, event_times

dev = torch.device("cuda:0"); lp.config.check_inputs = False


def tile_perm(H, W, th, tw):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    key = ((ys // th) * (W // tw) + (xs // tw)) * (th * tw) + (ys % th) * tw + (xs % tw)
    return torch.argsort(key.reshape(-1))


for name, H, W in (("cfg2", 256, 256), ("cfg3", 256, 256)):
    for th, tw in ((1, 32), (2, 16), (4, 8), (8, 4), (4, 4), (2, 8), (8, 8)):
        wl = RendererWorkload(name, 0, dev, None, _lib.LP_KERNEL_AUTO) if name == "cfg2" else SplatterWorkload(0, dev, None)
        perm = tile_perm(H, W, th, tw).to(dev)
        r = wl.rays
        for f in ("directions", "origins", "near", "far", "grid_idx"):
            setattr(r, f, getattr(r, f)[perm].contiguous())
        r.encoding = r.encoding.detach()[perm].contiguous().requires_grad_(True)
        if name == "cfg2":
            wl.up = [u[perm].contiguous() for u in wl.up]
        for _ in range(5):
            wl.step()
        f, b = event_times(wl, 10)
        print(json.dumps({"workload": name, "tile": f"{th}x{tw}", "fwd_ms": round(f, 4), "bwd_ms": round(b, 4),
                          "Mrays_per_s": round(wl.n_rays / (f + b) / 1e3, 3)}), flush=True)
        del wl
        torch.cuda.empty_cache()
