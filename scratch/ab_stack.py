import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(True)
variants = [(p, l) for p in (1, 0) for l in (22, 16, 12, 8, 4, 2)]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        ctx.set_option("raygen_pregen", v[0]); ctx.set_option("lds_stack_levels", v[1])
        ctx.kernel_time("raygen", reset=True)
        for i in range(2, 10): loop.frame(i)
        ms, n = ctx.kernel_time("raygen", reset=True)
        res[v].append(ms / n)
ctx.set_ray_statistics(True)
for v in variants:
    ctx.set_option("raygen_pregen", v[0]); ctx.set_option("lds_stack_levels", v[1])
    loop.frame(5); torch.cuda.synchronize()
    print(f"pregen {v[0]} levels {v[1]:2d}: {np.median(res[v]):.4f} ms   overflows {ctx.ray_statistics()['stack_overflows']}")
