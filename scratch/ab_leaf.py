import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
scene = scenes.sponza_proc()
for leaf in (4, 3, 2, 1):
    c = lib.Context(8, 8); c.set_option("bvh_leaf_triangles", leaf); c.close()     # global builder knob
    loop = HybridFrameLoop(scene, 1920, 1080, 12, reflections=False)
    ctx = loop.ctx
    ctx.set_kernel_timing(True)
    for i in range(2, 6): loop.frame(i)
    ctx.kernel_time("raygen", reset=True)
    for r in range(3):
        for i in range(2, 10): loop.frame(i)
    ms, n = ctx.kernel_time("raygen")
    ctx.set_ray_statistics(True); loop.frame(4); torch.cuda.synchronize()
    ts = ctx.traversal_statistics(); rs = ctx.ray_statistics(); b = ctx.bvh_statistics()
    r = rs['unique_rays']
    print(f"leaf<= {leaf}: {ms/n:.4f} ms nodes/ray {ts['node_visits']/r:.1f} leaves/ray {ts['leaf_visits']/r:.2f} tris/ray {ts['triangle_tests']/r:.2f} bvh nodes {b['nodes']} depth {b['max_depth']}")
    loop.close()
