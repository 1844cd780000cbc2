"""Does a SPACE-FILLING order of an image's rays help the run-merged atomic walks?  scripts/bench_tile_order.py (round 3) put th x tw pixel
tiles into a wave in row-major order inside the tile -- vertical neighbours are then tw lanes apart and never merge.  Here consecutive
lanes are always pixel neighbours: a 2-row zigzag (column by column through two image rows), a 4-row snake, and Morton order inside
8 x 4 tiles.  cfg-2 workload (Renderer) and cfg-3 workload (Splatter), forward + backward, HIP events."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from bench import RendererWorkload, SplatterWorkload, event_times

dev = torch.device("cuda:0"); lp.config.check_inputs = False


def order(H, W, kind):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    if kind == "row_major":
        key = ys * W + xs
    elif kind == "zigzag2":      # pairs of rows, column by column: (y0,x),(y0+1,x),(y0+1,x+1),(y0,x+1),...
        sub = torch.where(xs % 2 == 0, ys % 2, 1 - ys % 2)
        key = (ys // 2) * (2 * W) + xs * 2 + sub
    elif kind == "snake4":       # four rows, column by column, alternating direction
        sub = torch.where(xs % 2 == 0, ys % 4, 3 - ys % 4)
        key = (ys // 4) * (4 * W) + xs * 4 + sub
    elif kind == "morton8x4":    # 8 wide x 4 high tiles (32 rays = one wave), Morton order inside
        tx, ty = xs % 8, ys % 4
        m = (tx & 1) | ((ty & 1) << 1) | ((tx & 2) << 1) | ((ty & 2) << 2) | ((tx & 4) << 2)
        key = ((ys // 4) * (W // 8) + xs // 8) * 32 + m
    return torch.argsort(key.reshape(-1))


for name, H, W in (("cfg3", 256, 256), ("cfg2", 256, 256)):
    for kind in ("row_major", "zigzag2", "snake4", "morton8x4"):
        wl = RendererWorkload(name, 0, dev, None, _lib.LP_KERNEL_AUTO) if name == "cfg2" else SplatterWorkload(0, dev, None)
        perm = order(H, W, kind).to(dev)
        r = wl.rays
        for f in ("directions", "origins", "near", "far", "grid_idx"):
            setattr(r, f, getattr(r, f)[perm].contiguous())
        r.encoding = r.encoding.detach()[perm].contiguous().requires_grad_(True)
        if name == "cfg2":
            wl.up = [u[perm].contiguous() for u in wl.up]
        for _ in range(5):
            wl.step()
        f, b = event_times(wl, 10)
        print(json.dumps({"workload": name, "order": kind, "fwd_ms": round(f, 4), "bwd_ms": round(b, 4),
                          "Mrays_per_s": round(wl.n_rays / (f + b) / 1e3, 3)}), flush=True)
        del wl
        torch.cuda.empty_cache()
