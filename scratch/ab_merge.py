"""A/B of the in-block tail merge (option "raygen_merge") at 2 and 4 waves per block: raygen kernel time, wave trips, identical images."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
for name in ("sponza_proc", "bistro_proc"):
    for shadow, ao in ((True, 2), (False, 2), (True, 0)):
        loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12, shadow=shadow, ao_spp=ao, reflections=False, denoise=True)
        ctx = loop.ctx
        ref = None
        for wv, mg in ((2, 0), (2, 1), (4, 0), (4, 1)):
            ctx.set_option("raygen_waves_per_block", wv)
            ctx.set_option("raygen_merge", mg)
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
            ts = ctx.traversal_statistics()
            ctx.set_ray_statistics(False)
            loop.frame(5); torch.cuda.synchronize()
            img = ctx.download(lib.RAYTRACED)
            if ref is None: ref = img
            print(f"{name} shadow={shadow} ao={ao} waves/block {wv} merge {mg}: {ms/8*1e3:.1f} us, lanes {ts['active_lane_utilisation']:.3f}, "
                  f"wave trips {ts['wave_iterations']}, identical {np.array_equal(img, ref)}", flush=True)
        loop.close()
