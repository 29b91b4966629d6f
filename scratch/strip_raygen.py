"""Thin-strip ray tracing (the 195 rows one of 8 GPUs traces at 1080p): kernel time vs block shape / shared-tile mode."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 16)
ctx = loop.ctx
for n in (8, 4, 2, 1):
    plan = tiling.make_plan(H, n, n // 2, loop.max_motion_rows)
    ctx.set_strip(plan.row_begin, plan.row_end, plan.overlap, plan.halo)
    print(plan)
    ctx.set_option("trace_overlap", 1)
    for shared, waves, thr in ((8, 2, 16), (4, 2, 16), (2, 2, 16), (4, 1, 16), (4, 4, 16), (0, 2, 16)):
        ctx.set_option("raygen_tile_rows", shared); ctx.set_option("raygen_waves_per_block", waves); ctx.set_option("refill_threshold", thr)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 15): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("raygen")
        ctx.set_kernel_timing(False)
        print(f"N={n} tile_rows={shared} waves={waves} refill={thr}: raygen {ms / k * 1e3:.1f} us", flush=True)
loop.close()
