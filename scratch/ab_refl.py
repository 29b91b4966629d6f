"""Mirror-ray and raytraced-path kernel times (A-B across builds)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    for refl in (1, 2):
        loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12, reflections=refl)
        ctx = loop.ctx
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen", "reflection"]); ctx.kernel_time("raygen", reset=True); ctx.kernel_time("reflection", reset=True)
        for i in range(3, 11): loop.frame(i)
        ctx.synchronize()
        (a, k), (b, k2) = ctx.kernel_time("raygen"), ctx.kernel_time("reflection")
        print(f"{name} bounces {refl}: raygen {a/k*1e3:.1f} us, reflection {b/max(1,k2)*1e3:.1f} us", flush=True)
        loop.close()
