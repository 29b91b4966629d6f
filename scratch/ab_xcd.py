import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(True)
variants = [(w, x) for w in (1, 2, 4) for x in (0, 1)]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        ctx.set_option("raygen_waves_per_block", v[0]); ctx.set_option("xcd_aware", v[1])
        ctx.kernel_time("raygen", reset=True)
        for i in range(2, 10): loop.frame(i)
        ms, n = ctx.kernel_time("raygen", reset=True)
        res[v].append(ms / n)
for v in variants:
    print(f"waves/block {v[0]} xcd_aware {v[1]}: {np.median(res[v]):.4f} ms")
