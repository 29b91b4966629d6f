"""Kernel times of the busiest rank's rectangle at N = 1, 2, 4, 8 (virtual tiles on one GPU, event pairs on every launch): where a thin tile's frame goes.
usage: python scratch/tile_kernels.py [config2|config4]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
scene_name, W, H, ao, refl = {"config2": ("sponza_proc", 1920, 1080, 2, 0), "config4": ("bistro_proc", 1920, 1080, 2, 1)}[name]
loop = HybridFrameLoop(getattr(scenes, scene_name)(), W, H, 12, ao_spp=ao, reflections=refl)
ctx = loop.ctx
area = lambda p: (p.computed_rect()[1] - p.computed_rect()[0]) * (p.computed_rect()[3] - p.computed_rect()[2])
for n in (1, 2, 4, 8):
    plans = [tiling.make_tile_plan(W, H, n, r, loop.max_motion_rows, loop.max_motion_cols) for r in range(n)]
    p = max(plans, key=area)
    ctx.set_tile(p.col_begin, p.col_end, p.row_begin, p.row_end, p.overlap, p.halo_rows, p.halo_cols)
    ctx.set_option("trace_overlap", 1 if n > 1 else 0); ctx.set_option("strip_shrink_overlap", 1 if n > 1 else 0)
    for i in range(3): loop.frame(i)
    torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 11): loop.frame(i)
    torch.cuda.synchronize(); ctx.synchronize()
    frame_us = (time.perf_counter() - t0) / 8 * 1e6
    kinds = ["raygen", "svgf_temporal", "svgf_atrous", "svgf_atrous_async"] + (["reflection"] if refl else [])
    ctx.set_kernel_timing(kinds)
    for k in kinds: ctx.kernel_time(k, reset=True)
    for i in range(3, 11): loop.frame(i)
    torch.cuda.synchronize(); ctx.synchronize()
    out = {"config": name, "n": n, "computed_pixels": area(p), "share_of_frame_pixels": round(area(p) / (W * H), 4), "frame_us_untimed": round(frame_us, 1)}
    for k in kinds:
        ms, cnt = ctx.kernel_time(k)
        out[k] = {"us_per_launch": round(ms / max(1, cnt) * 1e3, 2), "launches_per_frame": cnt / 8}
    ctx.set_kernel_timing(False)
    print(json.dumps(out), flush=True)
loop.close()
