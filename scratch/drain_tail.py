"""How much of the ray-tracing kernel's trips are made with only a few rays of the wave left (the tail of each tile's drain)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for k, v in opts: ctx.set_option(k, int(v))
    for i in range(3): loop.frame(i)
    ctx.set_ray_statistics(True); loop.frame(5); ctx.synchronize()
    ts, ps, cy = ctx.traversal_statistics(), ctx.drain_statistics(), ctx.traversal_cycles()
    T = ts["wave_iterations"]
    n = max(1, ts["rays"]) if "rays" in ts else 1
    print(f"{name} {opts}: lane utilisation {(ts['node_visits'] + ts['triangle_tests']) / (64.0 * T):.3f}, node visits {ts['node_visits']}, trips {T}, drain {cy['drain_iterations']/T:.3f}, with <= 16 rays left {ps['drain_trips_le16']/T:.3f}, <= 8 {ps['drain_trips_le8']/T:.3f}, <= 4 {ps['drain_trips_le4']/T:.3f}", flush=True)
    loop.close()
