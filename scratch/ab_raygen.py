import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
scene = scenes.sponza_proc()
W, H = 1920, 1080
loop = HybridFrameLoop(scene, W, H, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(True)
variants = [(0, 16, 6), (1, 1, 6), (1, 8, 6), (1, 16, 6), (1, 24, 6), (1, 32, 6), (1, 48, 6), (1, 64, 6)]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        ctx.set_option("raygen_variant", v[0]); ctx.set_option("refill_threshold", v[1]); ctx.set_option("raygen_blocks_per_cu", v[2])
        ctx.kernel_time("raygen", reset=True)
        for i in range(2, 10):
            loop.frame(i)
        ms, n = ctx.kernel_time("raygen", reset=True)
        res[v].append(ms / n)
rays = loop.rays_in_frame(4)
for v in variants:
    r = res[v]
    print(f"variant {v}: median {np.median(r):.4f} ms  min {min(r):.4f}  -> {rays/np.median(r)/1e6:.2f} Grays/s")
