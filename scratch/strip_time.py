"""Compute-only frame time of ONE rank's share of the 1080p frame (the busiest rank of N), no exchanges: the lower bound a rank of an
N-GPU run needs per frame, with the overlap margin's rays traced locally.  Row strips and the planner's screen tiles side by side.
usage: python scratch/strip_time.py [frames_in_flight]"""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop

W, H = 1920, 1080
scene = scenes.sponza_proc()
args = [a for a in sys.argv[1:] if "=" not in a]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
FIF = int(args[0]) if args else 1
loop = HybridFrameLoop(scene, W, H, 24, shadow=True, ao_spp=2, reflections=False, denoise=True, frames_in_flight=FIF)
for k, v in opts: loop.ctx.set_option(k, int(v))
base = None
for n in (1, 2, 4, 8):
    for grid in (("strips", None) if n > 1 else ("strips",)):
        plans = [tiling.make_tile_plan(W, H, n, r, loop.max_motion_rows, loop.max_motion_cols, grid=grid) for r in range(n)]
        area = lambda p: (p.computed_rect()[1] - p.computed_rect()[0]) * (p.computed_rect()[3] - p.computed_rect()[2])
        plan = max(plans, key=area)                       # the busiest rank
        loop.ctx.set_tile(plan.col_begin, plan.col_end, plan.row_begin, plan.row_end, plan.overlap, plan.halo_rows, plan.halo_cols)
        loop.ctx.set_option("trace_overlap", 1 if n > 1 else 0)
        loop.ctx.set_option("strip_shrink_overlap", 1 if n > 1 else 0)
        for i in range(4): loop.frame(i)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(4, 24): loop.frame(i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        loop.ctx.set_kernel_timing(["raygen", "svgf_temporal", "svgf_atrous", "svgf_atrous_async", "blit"])
        for k in ("raygen", "svgf_temporal", "svgf_atrous", "svgf_atrous_async", "blit"): loop.ctx.kernel_time(k, reset=True)
        for i in range(4, 24): loop.frame(i)
        torch.cuda.synchronize()
        kt = {k: loop.ctx.kernel_time(k) for k in ("raygen", "svgf_temporal", "svgf_atrous", "svgf_atrous_async", "blit")}
        loop.ctx.set_kernel_timing(False)
        ms = float(np.median(ts))
        base = ms if n == 1 else base
        c = plan.computed_rect()
        print(f"frames_in_flight {FIF} N={n} grid {plan.grid_rows}x{plan.grid_cols} owned {plan.col_end - plan.col_begin}x{plan.row_end - plan.row_begin} computed {c[1] - c[0]}x{c[3] - c[2]} "
              f"(+{100.0 * area(plan) * n / (W * H) - 100.0:.0f} %): {ms:.4f} ms/frame = {100.0 * base / (n * ms):.0f} % of linear  " +
              " ".join(f"{k} {v[0]/max(1,v[1])*1e3:.1f}us x{v[1]}" for k, v in kt.items()), flush=True)
loop.close()
