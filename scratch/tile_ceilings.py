"""r4: compute-only frame time of the BUSIEST rank's rectangle at N = 1, 2, 4, 8 for the configurations where the design claims
to scale, on one GPU with virtual tiles (no exchanges): config 3 (sponza_proc 4K, 4 AO samples), config 4 (bistro_proc 1080p, full hybrid),
config 5 (bistro_proc 4K, 16 AO samples, two bounces); the planner's grid and row strips, one stream and two frames in flight -- and, for the
planner's grid at N = 8, every rank's rectangle (the spread between sky tiles and street tiles).
usage: python scratch/tile_ceilings.py config3|config4|config5|config2 [frames_in_flight] [option=value ...]"""
import sys, time, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop

CONFIGS = {"config2": ("sponza_proc", 1920, 1080, 2, 0), "config3": ("sponza_proc", 3840, 2160, 4, 0), "config4": ("bistro_proc", 1920, 1080, 2, 1),
           "config5": ("bistro_proc", 3840, 2160, 16, 2)}
args = [a for a in sys.argv[1:] if "=" not in a]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
name = args[0]
FIF = int(args[1]) if len(args) > 1 else 1
scene_name, W, H, ao, refl = CONFIGS[name]
scene = getattr(scenes, scene_name)()
NF = 12
loop = HybridFrameLoop(scene, W, H, NF, shadow=True, ao_spp=ao, reflections=refl, denoise=True, frames_in_flight=FIF)
for k, v in opts: loop.ctx.set_option(k, int(v))

def measure(plan, n, reps=3, frames=8):
    loop.ctx.set_tile(plan.col_begin, plan.col_end, plan.row_begin, plan.row_end, plan.overlap, plan.halo_rows, plan.halo_cols)
    loop.ctx.set_option("trace_overlap", 1 if n > 1 else 0)
    loop.ctx.set_option("strip_shrink_overlap", 1 if n > 1 else 0)
    for i in range(3): loop.frame(i)
    torch.cuda.synchronize(); loop.ctx.synchronize()
    ts = []
    for rep in range(reps):
        t0 = time.perf_counter()
        for i in range(3, 3 + frames): loop.frame(i)
        torch.cuda.synchronize(); loop.ctx.synchronize()
        ts.append((time.perf_counter() - t0) / frames * 1e3)
    return float(np.median(ts))

area = lambda p: (p.computed_rect()[1] - p.computed_rect()[0]) * (p.computed_rect()[3] - p.computed_rect()[2])
base = None
for n in (1, 2, 4, 8):
    for grid in (("strips", None) if n > 1 else ("strips",)):
        try:
            plans = [tiling.make_tile_plan(W, H, n, r, loop.max_motion_rows, loop.max_motion_cols, grid=grid) for r in range(n)]
        except ValueError as e:
            print(f"{name} N={n} grid {grid}: {e}", flush=True); continue
        plan = max(plans, key=area)                       # the rank that computes the most pixels
        ms = measure(plan, n)
        base = ms if n == 1 else base
        c = plan.computed_rect()
        line = {"config": name, "frames_in_flight": FIF, "n": n, "grid": f"{plan.grid_rows}x{plan.grid_cols}", "owned": [plan.col_end - plan.col_begin, plan.row_end - plan.row_begin],
                "computed": [c[1] - c[0], c[3] - c[2]], "extra_pixels_pct": round(100.0 * area(plan) * n / (W * H) - 100.0, 1), "ms_per_frame": round(ms, 4),
                "share_of_linear_pct": round(100.0 * base / (n * ms), 1)}
        if n == 8 and grid is None and FIF == 1:          # every rank's rectangle of the planner's grid: the balance between tiles
            per = [round(measure(p, n, reps=2, frames=6), 4) for p in plans]
            line["per_rank_ms"] = per
            line["max_over_mean"] = round(max(per) / (sum(per) / len(per)), 3)
            line["share_of_linear_by_slowest_rank_pct"] = round(100.0 * base / (n * max(per)), 1)
        print(json.dumps(line), flush=True)
loop.close()
