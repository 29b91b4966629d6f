import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(["raygen"])
def t(y0, y1):
    ctx.set_strip(y0, y1, 0, 0)
    for i in range(2, 5): loop.frame(i)
    ctx.kernel_time("raygen", reset=True)
    for r in range(2):
        for i in range(2, 10): loop.frame(i)
    ms, n = ctx.kernel_time("raygen", reset=True)
    return ms / n
for shared, waves in ((0, 2), (1, 2), (1, 3), (1, 4), (0, 1)):
    ctx.set_option("raygen_shared_tile", shared); ctx.set_option("raygen_waves_per_block", waves)
    full = t(0, 1080)
    eighth = [t(k * 135, (k + 1) * 135) for k in (0, 3, 7)]
    print(f"shared {shared} waves {waves}: full {full:.4f} ms; 1/8 strips {[round(x, 3) for x in eighth]}")
