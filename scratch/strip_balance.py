"""Compute-only frame time of every rank's strip at N = 2, 4, 8 (equal strips): how unbalanced are they?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
scene = getattr(scenes, sys.argv[1] if len(sys.argv) > 1 else "sponza_proc")()
loop = HybridFrameLoop(scene, W, H, 24, shadow=True, ao_spp=2, reflections=False, denoise=True)
loop.ctx.set_option("strip_shrink_overlap", 1)
for n in (2, 4, 8):
    ts = []
    for r in range(n):
        plan = tiling.make_plan(H, n, r, loop.max_motion_rows)
        loop.ctx.set_strip(plan.row_begin, plan.row_end, plan.overlap, plan.halo)
        loop.ctx.set_option("trace_overlap", 1)
        for i in range(4): loop.frame(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(3):
            for i in range(4, 24): loop.frame(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 60 * 1e3)
    print(f"N={n}: " + " ".join(f"{t:.3f}" for t in ts) + f"  max {max(ts):.3f} mean {np.mean(ts):.3f} max/mean {max(ts)/np.mean(ts):.2f}", flush=True)
loop.close()
