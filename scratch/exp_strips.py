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
    for i in range(2, 6): loop.frame(i)
    ctx.kernel_time("raygen", reset=True)
    for r in range(2):
        for i in range(2, 10): loop.frame(i)
    ms, n = ctx.kernel_time("raygen", reset=True)
    return ms / n
full = t(0, 1080)
print("full", round(full, 4))
for parts in (2, 4, 8, 16):
    ts = [t(k * 1080 // parts, (k + 1) * 1080 // parts) for k in range(parts)]
    print(parts, "strips: sum", round(sum(ts), 4), "each", [round(x, 3) for x in ts])
