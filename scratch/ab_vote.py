import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12, reflections=False)
ctx = loop.ctx
ctx.set_kernel_timing(["raygen"])
variants = [(v, t) for v in (0, 1) for t in (4, 8, 16, 32)]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        ctx.set_option("traversal_vote", v[0]); ctx.set_option("refill_threshold", v[1])
        ctx.kernel_time("raygen", reset=True)
        for i in range(2, 10): loop.frame(i)
        ms, n = ctx.kernel_time("raygen", reset=True)
        res[v].append(ms / n)
ctx.set_ray_statistics(True)
for v in variants:
    ctx.set_option("traversal_vote", v[0]); ctx.set_option("refill_threshold", v[1])
    loop.frame(5); torch.cuda.synchronize()
    ts = ctx.traversal_statistics()
    print(f"vote {v[0]} refill {v[1]}: {np.median(res[v]):.4f} ms  lane utilisation {ts['active_lane_utilisation']:.3f}")
