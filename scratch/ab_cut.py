import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12)
    ctx = loop.ctx
    ref = None
    for cut in (0, 1):
        ctx.set_option("raygen_cut", cut)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
        ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
        rs, ts = ctx.ray_statistics(), ctx.traversal_statistics()
        ctx.set_ray_statistics(False)
        img = ctx.download(lib.RAYTRACED)
        if ref is None: ref = img
        print(f"{name} cut {cut}: {ms/k*1e3:.1f} us, node trips/ray {ts['node_visits']/max(1,rs['unique_rays']):.2f}, util {ts['active_lane_utilisation']:.3f}, identical {np.array_equal(img, ref)}, overflows {rs['stack_overflows']}", flush=True)
    loop.close()
