import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(True)
variants = [(w, t) for w in (4, 2) for t in (0, 31, 63, 127, 255, 511, 1023)]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        ctx.set_option("raygen_waves_per_block", v[0]); ctx.set_option("lds_top_nodes", v[1])
        ctx.kernel_time("raygen", reset=True)
        for i in range(2, 10): loop.frame(i)
        ms, n = ctx.kernel_time("raygen", reset=True)
        res[v].append(ms / n)
for v in variants:
    print(f"waves/block {v[0]} top nodes {v[1]:4d}: {np.median(res[v]):.4f} ms")
