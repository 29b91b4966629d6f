"""A/B of the tail hand-off (option "raygen_tail" = hand off once at most that many rays of a tile are left): raygen time
(main + tail kernel, HIP events around both), identical images."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
CASES = ((True, 2),) if os.environ.get("AB_SHORT") else ((True, 2), (True, 0), (False, 4))
for name in ("sponza_proc", "bistro_proc"):
    for shadow, ao in CASES:
        loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12, shadow=shadow, ao_spp=ao, reflections=False, denoise=True)
        ctx = loop.ctx
        ref = None
        for tail in (0, 4, 8, 12, 16):
            ctx.set_option("raygen_tail", tail)
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            loop.frame(5); torch.cuda.synchronize()
            img = ctx.download(lib.RAYTRACED)
            if ref is None: ref = img
            print(f"{name} shadow={shadow} ao={ao} tail {tail}: {ms/8*1e3:.1f} us per frame over {k//8} launches, identical {np.array_equal(img, ref)}", flush=True)
        loop.close()
