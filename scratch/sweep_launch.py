"""Re-sweep of the queue kernel's launch options on the current kernel: ray-tracing kernel time (min of 3 x 8 frames, the side stream off), bit-identity.
usage: python scratch/sweep_launch.py [scene ...]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
arms = [dict(), dict(xcd_aware=1), dict(raygen_waves_per_block=4), dict(raygen_waves_per_block=1), dict(refill_threshold=8), dict(refill_threshold=24), dict(refill_threshold=32),
        dict(raygen_early_exit=3), dict(raygen_early_exit=5), dict(lds_stack_levels=6), dict(lds_stack_levels=7), dict(cut_expand=1), dict(shadow_last=0), dict()]
defaults = dict(xcd_aware=0, raygen_waves_per_block=2, refill_threshold=16, raygen_early_exit=4, lds_stack_levels=8, cut_expand=0, shadow_last=1)
for name in (sys.argv[1:] or ["sponza_proc", "bistro_proc"]):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12)
    ctx = loop.ctx
    ctx.set_option("svgf_async_unread", 0)
    ref = None
    for arm in arms:
        for k, v in defaults.items(): ctx.set_option(k, v)
        for k, v in arm.items(): ctx.set_option(k, v)
        times = []
        for rep in range(3):
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            times.append(ms / 8 * 1e3)
        loop.frame(5); torch.cuda.synchronize()
        h = hashlib.md5(ctx.download(lib.RAYTRACED).tobytes()).hexdigest()[:10]
        ref = ref or h
        print(f"{name} {arm}: raygen {min(times):.1f} us {[round(t, 1) for t in times]} identical {h == ref}", flush=True)
    loop.close()
