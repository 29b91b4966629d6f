import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(True)
for mode in (0, 1, 2, 0):
    ctx.set_option("debug_leaf_mode", mode)
    for i in range(2, 6): loop.frame(i)
    ctx.kernel_time("raygen", reset=True)
    for r in range(3):
        for i in range(2, 10): loop.frame(i)
    ms, n = ctx.kernel_time("raygen")
    ctx.set_ray_statistics(True); loop.frame(4); torch.cuda.synchronize()
    ts = ctx.traversal_statistics(); rs = ctx.ray_statistics(); ctx.set_ray_statistics(False)
    r = rs['unique_rays']
    print(f"leaf mode {mode}: {ms/n:.4f} ms nodes/ray {ts['node_visits']/r:.1f} leaves/ray {ts['leaf_visits']/r:.2f} iters {ts['wave_iterations']}")
