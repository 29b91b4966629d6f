"""Leaf size of the tree ("bvh_leaf_triangles") against the any-hit launch (VHR_KERNEL=reflection: the mirror-ray launch, with reflection_async 0): build, nodes, visits and tests per ray, time (arms interleaved)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    scene = getattr(scenes, name)()
    kernel = os.environ.get("VHR_KERNEL", "raygen")
    loop = HybridFrameLoop(scene, 1920, 1080, 12, reflections=(kernel == "reflection"))
    ctx = loop.ctx
    ctx.set_option("reflection_async", 0)
    times = {2: [], 3: []}
    info = {}
    for rep in range(5):
        for leaf in (2, 3):
            ctx.set_option("bvh_leaf_triangles", leaf)
            ctx.upload_scene(scene)
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing([kernel]); ctx.kernel_time(kernel, reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time(kernel); ctx.set_kernel_timing(False)
            times[leaf].append(ms / 8 * 1e3)
            if rep == 0:
                ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
                rs, ts = ctx.ray_statistics(), ctx.traversal_statistics(); ctx.set_ray_statistics(False)
                n = max(1, rs["unique_rays"])
                info[leaf] = (ctx.bvh_statistics()["nodes"], ts["node_visits"] / n, ts["triangle_tests"] / n)
    for leaf in (2, 3):
        print(f"{name} leaf {leaf}: nodes {info[leaf][0]}, visits/ray {info[leaf][1]:.2f}, tests/ray {info[leaf][2]:.2f}, {kernel} {min(times[leaf]):.1f} us {[round(t, 1) for t in times[leaf]]}", flush=True)
    loop.close()
