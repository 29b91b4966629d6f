"""Leaf size of the host BVH build (option bvh_leaf_triangles, applies to the next upload) on the round-2 kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    scene = getattr(scenes, name)()
    loop = HybridFrameLoop(scene, 1920, 1080, 12)
    ctx = loop.ctx
    ref = None
    for leaf in (4, 4, 3, 2, 1, 4):
        ctx.set_option("bvh_leaf_triangles", leaf)
        ctx.upload_scene(scene)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
        ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
        rs, ts = ctx.ray_statistics(), ctx.traversal_statistics(); ctx.set_ray_statistics(False)
        loop.frame(5); torch.cuda.synchronize()
        img = ctx.download(lib.RAYTRACED)
        if ref is None: ref = img
        n = max(1, rs["unique_rays"])
        print(f"{name} leaf {leaf}: {ms / 8 * 1e3:.1f} us, nodes {ctx.bvh_statistics()['nodes']}, node visits/ray {ts['node_visits'] / n:.2f}, tri tests/ray {ts['triangle_tests'] / n:.2f}, "
              f"wave trips {ts['wave_iterations']}, identical {np.array_equal(img, ref)}", flush=True)
    loop.close()
