import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 36)
ctx = loop.ctx
for stamps in (1, 0, 1, 0):
    ctx.set_option("pass_timestamps", stamps)
    for i in range(4): loop.frame(i)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(4, 36): loop.frame(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 32 * 1e3)
    print(f"pass_timestamps {stamps}: {np.median(ts):.4f} ms/frame", flush=True)
loop.close()
