import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, 1920, 1080, 24)
def run(tag):
    loop.ctx.set_kernel_timing(["svgf_atrous"])
    loop.ctx.kernel_time("svgf_atrous", reset=True)
    for i in range(24): loop.frame(i)
    torch.cuda.synchronize()
    t, n = loop.ctx.kernel_time("svgf_atrous")
    print(f"{tag}: {t / n * 1e3:.2f} us/launch ({n} launches)", flush=True)
    loop.ctx.set_kernel_timing(False)
for v in (3, 4, 3, 4):
    loop.ctx.set_option("atrous_variant", v)
    run(f"atrous_variant {v}")
for b in (2, 3, 4, 5, 6, 8):
    for x in (0, 1):
        loop.ctx.set_option("atrous_blocks_per_cu", b); loop.ctx.set_option("atrous_xcd_aware", x)
        run(f"variant 4 blocks/CU {b} xcd {x}")
loop.close()
