"""Where the traversal time goes by ray kind: shadow only / AO only / both (kernel time, node visits, lane utilisation)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
scene = scenes.sponza_proc()
for shadow, ao in ((True, 0), (False, 2), (True, 2), (False, 1), (True, 4)):
    loop = HybridFrameLoop(scene, 1920, 1080, 12, shadow=shadow, ao_spp=ao, reflections=False, denoise=True)
    ctx = loop.ctx
    for i in range(4): loop.frame(i)
    ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
    for i in range(4, 12): loop.frame(i)
    torch.cuda.synchronize()
    t, n = ctx.kernel_time("raygen")
    ctx.set_kernel_timing(False)
    ctx.set_ray_statistics(True)
    loop.frame(5); torch.cuda.synchronize()
    rs, ts = ctx.ray_statistics(), ctx.traversal_statistics()
    r = rs["unique_rays"]
    print(f"shadow={shadow} ao={ao}: {t/n*1e3:.1f} us, rays {r/1e6:.2f} M, {r/(t/n)/1e6:.0f} Mrays/s... nodes/ray {ts['node_visits']/r:.1f} leaves/ray {ts['leaf_visits']/r:.2f} tris/ray {ts['triangle_tests']/r:.2f} util {ts['active_lane_utilisation']:.3f}", flush=True)
    loop.close()
