"""A/B of the mirror-ray work-queue kernel's runtime knobs (all settings compute identical images)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    scene = getattr(scenes, name)()
    loop = HybridFrameLoop(scene, 1920, 1080, 12, shadow=True, ao_spp=2, reflections=True, denoise=True)
    ctx = loop.ctx
    def t(**opts):
        for k, v in opts.items(): ctx.set_option(k, v)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["reflection"]); ctx.kernel_time("reflection", reset=True)
        for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time("reflection")
        ctx.set_kernel_timing(False)
        return ms / n * 1e3
    base = dict(reflection_variant=1, refill_threshold=16, raygen_early_exit=4, lds_stack_levels=8)
    print(name, "variant 0:", round(t(reflection_variant=0), 1), "us; default:", round(t(**base), 1), flush=True)
    for key, vals in (("refill_threshold", (1, 8, 24, 32, 48, 64)), ("raygen_early_exit", (0, 2, 6, 8)), ("lds_stack_levels", (4, 12, 16, 24))):
        res = []
        for v in vals:
            o = dict(base); o[key] = v
            res.append((v, round(t(**o), 1)))
        print("  ", key, res, flush=True)
    t(**base)
    loop.close()
