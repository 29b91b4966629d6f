import sys, os, hashlib
sys.path.insert(0, '/root/repo')
from vulkanhybridrenderer_amd import lib
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 12)
ctx = loop.ctx
ctx.set_option("svgf_async_unread", 0)
VARS = (0, 1); res = {v: [] for v in VARS}
for rep in range(4):
    for v in VARS:
        ctx.set_option("temporal_variant", v)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["svgf_temporal"]); ctx.kernel_time("svgf_temporal", reset=True)
        for r in range(2):
            for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        tms, tk = ctx.kernel_time("svgf_temporal"); ctx.set_kernel_timing(False)
        res[v].append(tms / tk * 1e3)
for v in VARS: print(f"temporal_variant {v}: {min(res[v]):.2f} us ({[round(t, 2) for t in res[v]]})", flush=True)
loop.close()
