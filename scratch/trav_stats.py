import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 8, reflections=False)
ctx = loop.ctx
ctx.set_ray_statistics(True)
for thr in (64, 24, 8, 1):
    ctx.set_option("refill_threshold", thr)
    loop.frame(5); torch.cuda.synchronize()
    rs, ts = ctx.ray_statistics(), ctx.traversal_statistics()
    r = rs["unique_rays"]
    print(f"thr {thr}: rays {r} nodes/ray {ts['node_visits']/r:.1f} leaves/ray {ts['leaf_visits']/r:.2f} tris/ray {ts['triangle_tests']/r:.2f} util {ts['active_lane_utilisation']:.3f} wave_iters {ts['wave_iterations']}")
