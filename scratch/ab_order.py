"""A/B of the cut expansion (option "shadow_last"): raygen kernel time, node visits per ray, cut entries per tile, identical images."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
for name in ("sponza_proc", "bistro_proc"):
    for shadow, ao in ((True, 2), (True, 4)):
        loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12, shadow=shadow, ao_spp=ao, reflections=False, denoise=True)
        ctx = loop.ctx
        ref = None
        for ex in (0, 1):
            ctx.set_option("shadow_last", ex)
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
            rs, ts, ps, cy = ctx.ray_statistics(), ctx.traversal_statistics(), ctx.packet_statistics(), ctx.traversal_cycles()
            ctx.set_ray_statistics(False)
            img = ctx.download(lib.RAYTRACED)
            if ref is None: ref = img
            print(f"{name} shadow={shadow} ao={ao} shadow_last {ex}: {ms/8*1e3:.1f} us, cut entries/tile {ps['cut_entries']/max(1,cy['waves']):.1f}, "
                  f"node visits/ray {ts['node_visits']/max(1,rs[0] if not isinstance(rs, dict) else rs.get('unique_rays',1)):.2f}, tri tests/ray {ts['triangle_tests']/max(1,rs[0] if not isinstance(rs, dict) else rs.get('unique_rays',1)):.2f}, lanes {ts['active_lane_utilisation']:.3f}, "
                  f"wave trips {ts['wave_iterations']}, identical {np.array_equal(img, ref)}", flush=True)
        loop.close()
