"""Compute-only frame time of ONE rank's strip (middle rank of N), no exchanges: the lower bound a rank of an
N-GPU run needs per frame, with the overlap rows' rays traced locally (1) or assumed delivered by a neighbour (0)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop

W, H = 1920, 1080
scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, W, H, 24, shadow=True, ao_spp=2, reflections=False, denoise=True)
for n in (1, 2, 4, 8):
    for shared in (0, 1):
        for to in ((0, 1) if n > 1 else (0,)):
            r = n // 2
            plan = tiling.make_plan(H, n, r, loop.max_motion_rows)
            loop.ctx.set_strip(plan.row_begin, plan.row_end, plan.overlap, plan.halo)
            loop.ctx.set_option("trace_overlap", to)
            loop.ctx.set_option("raygen_shared_tile", shared)
            for i in range(4): loop.frame(i)
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                t0 = time.perf_counter()
                for i in range(4, 24): loop.frame(i)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
            print(f"N={n} rank {r} rows {plan.rows}+{2*plan.overlap if n>1 else 0} shared_tile={shared} trace_overlap={to}: {np.median(ts):.4f} ms/frame", flush=True)
loop.close()
