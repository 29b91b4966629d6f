"""What the half-precision nodes' wider boxes cost: node visits and triangle tests per ray on the 32-byte (compact_nodes 1) and the 48-byte nodes (0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for compact in (1, 0):
        ctx.set_option("compact_nodes", compact)
        for i in range(3): loop.frame(i)
        ctx.set_ray_statistics(True); loop.frame(5); ctx.synchronize()
        ts, rs = ctx.traversal_statistics(), ctx.ray_statistics()
        ctx.set_ray_statistics(False)
        n = max(1, rs["unique_rays"])
        print(f"{name} compact_nodes {compact}: node visits/ray {ts['node_visits'] / n:.3f}, leaf visits/ray {ts['leaf_visits'] / n:.3f}, triangle tests/ray {ts['triangle_tests'] / n:.3f}", flush=True)
    loop.close()
