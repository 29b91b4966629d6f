"""Node visits / triangle tests per ray of the any-hit launch on the 32-byte half-precision nodes against the 48-byte fp32-centre nodes (how much the coarser boxes cost in visits)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ["sponza_proc", "bistro_proc"]):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for compact in (1, 0):
        ctx.set_option("compact_nodes", compact)
        for i in range(3): loop.frame(i)
        ctx.set_ray_statistics(True); loop.frame(3); torch.cuda.synchronize(); ctx.synchronize()
        t, r = ctx.traversal_statistics(), ctx.ray_statistics()
        ctx.set_ray_statistics(False)
        print(json.dumps({"scene": name, "compact_nodes": compact, "visits_per_ray": round(t["node_visits"] / r["unique_rays"], 3), "tests_per_ray": round(t["triangle_tests"] / r["unique_rays"], 3),
                          "leaf_visits_per_ray": round(t["leaf_visits"] / r["unique_rays"], 3), "lanes": round(t["active_lane_utilisation"], 3)}), flush=True)
    loop.close()
