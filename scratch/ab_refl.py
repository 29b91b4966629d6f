"""Mirror-ray kernel (one and two bounces) on the 32-byte nodes (compact_nodes 1, default) against the 48-byte ones: kernel time, reflections image identical.
usage: python scratch/ab_refl.py [scene ...]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ("sponza_proc", "bistro_proc")):
    for refl in (1, 2):
        loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12, reflections=refl)
        ctx = loop.ctx
        ctx.set_option("svgf_async_unread", 0)
        ref = None
        for compact in (0, 1, 0, 1):
            ctx.set_option("compact_nodes", compact)
            times = []
            for rep in range(2):
                for i in range(3): loop.frame(i)
                ctx.set_kernel_timing(["raygen", "reflection"]); ctx.kernel_time("raygen", reset=True); ctx.kernel_time("reflection", reset=True)
                for i in range(3, 11): loop.frame(i)
                ctx.synchronize()
                (a, k), (b, k2) = ctx.kernel_time("raygen"), ctx.kernel_time("reflection")
                ctx.set_kernel_timing(False)
                times.append((a / k * 1e3, b / max(1, k2) * 1e3))
            loop.frame(5); ctx.synchronize()
            h = hashlib.md5(ctx.download(lib.REFLECTIONS).tobytes()).hexdigest()[:10]
            ref = ref or h
            print(f"{name} bounces {refl} compact_nodes {compact}: raygen {min(t[0] for t in times):.1f} us, reflection {min(t[1] for t in times):.1f} us, reflections identical {h == ref}", flush=True)
        loop.close()
