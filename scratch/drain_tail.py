"""How much of the ray-tracing kernel's trips are made with only a few rays of the wave left (the tail of each tile's drain)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for i in range(3): loop.frame(i)
    ctx.set_ray_statistics(True); loop.frame(5); ctx.synchronize()
    ts, ps, cy = ctx.traversal_statistics(), ctx.drain_statistics(), ctx.traversal_cycles()
    T = ts["wave_iterations"]
    print(f"{name}: trips {T}, drain {cy['drain_iterations']/T:.3f}, with <= 16 rays left {ps['drain_trips_le16']/T:.3f}, <= 8 {ps['drain_trips_le8']/T:.3f}, <= 4 {ps['drain_trips_le4']/T:.3f}", flush=True)
    loop.close()
