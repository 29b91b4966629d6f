import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, 1920, 1080, 24)
for v in (2, 3, 2, 3):
    loop.ctx.set_option("atrous_variant", v)
    loop.ctx.set_kernel_timing(["svgf_atrous"])
    loop.ctx.kernel_time("svgf_atrous", reset=True)
    for i in range(24): loop.frame(i)
    torch.cuda.synchronize()
    t, n = loop.ctx.kernel_time("svgf_atrous")
    print(f"atrous_variant {v}: {t / n * 1e3:.2f} us/launch ({n} launches)")
    loop.ctx.set_kernel_timing(False)
loop.close()
