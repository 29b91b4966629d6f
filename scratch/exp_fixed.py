import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, abi
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, W, H, 8, reflections=False)
ctx = loop.ctx
ctx.set_option("lds_top_nodes", 0); ctx.set_option("raygen_waves_per_block", 2)
ctx.set_kernel_timing(True)
for tmax in (1e4, 5.0, 1.0, 0.3, 0.1, 0.03, 1e-6):
    tp = loop.tp.copy(); tp["tmax"] = tmax; tp["ao_tmax"] = min(tmax, 5.0); ctx.set_trace_params(tp)
    for i in range(2, 6): loop.frame(i)
    ctx.kernel_time("raygen", reset=True)
    for r in range(3):
        for i in range(2, 8): loop.frame(i)
    ms, n = ctx.kernel_time("raygen")
    ctx.set_ray_statistics(True); loop.frame(4); torch.cuda.synchronize()
    ts = ctx.traversal_statistics(); rs = ctx.ray_statistics(); ctx.set_ray_statistics(False)
    r = max(1, rs['unique_rays'])
    print(f"tmax {tmax:8.2g}: {ms/n:.4f} ms nodes/ray {ts['node_visits']/r:5.1f} leaves/ray {ts['leaf_visits']/r:.2f} tris/ray {ts['triangle_tests']/r:.2f} outer iters {ts['wave_iterations']}")
