import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
for name in ("sponza_proc",):
    loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12)
    ctx = loop.ctx
    for i in range(3): loop.frame(i)
    ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
    print(ctx.ray_statistics()); print(ctx.traversal_statistics()); print(ctx.traversal_cycles())
    loop.close()
