"""One rank's frame of an N-tile decomposition under rocprofv3 --kernel-trace: run as
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 scratch/tile_frame_trace.py config2 8
then scratch/tile_frame_gaps.py <dir> prints, per frame, the kernels' busy time against the frame's span (what launch gaps cost a thin tile)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
CONFIGS = {"config2": ("sponza_proc", 1920, 1080, 2, 0), "config4": ("bistro_proc", 1920, 1080, 2, 1)}
name, n = sys.argv[1], int(sys.argv[2])
scene_name, W, H, ao, refl = CONFIGS[name]
loop = HybridFrameLoop(getattr(scenes, scene_name)(), W, H, 12, shadow=True, ao_spp=ao, reflections=refl, denoise=True)
plans = [tiling.make_tile_plan(W, H, n, r, loop.max_motion_rows, loop.max_motion_cols, grid=None) for r in range(n)]
area = lambda p: (p.computed_rect()[1] - p.computed_rect()[0]) * (p.computed_rect()[3] - p.computed_rect()[2])
plan = max(plans, key=area)
loop.ctx.set_tile(plan.col_begin, plan.col_end, plan.row_begin, plan.row_end, plan.overlap, plan.halo_rows, plan.halo_cols)
for k, v in (("trace_overlap", 1), ("strip_shrink_overlap", 1), ("reflection_async", 2)):
    loop.ctx.set_option(k, v if n > 1 else (0 if k != "reflection_async" else 1))
for kv in sys.argv[3:]:
    loop.ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for i in range(4): loop.frame(i)
torch.cuda.synchronize(); loop.ctx.synchronize()
t0 = time.perf_counter()
for i in range(4, 36): loop.frame(i % 12)
torch.cuda.synchronize(); loop.ctx.synchronize()
print(f"{name} N={n} computed {area(plan)} px: {(time.perf_counter() - t0) / 32 * 1e3:.4f} ms per frame by wall clock", flush=True)
loop.close()
