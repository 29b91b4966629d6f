"""A/B: the cut kernel on 64-byte fp32 nodes (centre / half extent) vs 32-byte half-precision nodes (option "compact_nodes")."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12)
    ctx = loop.ctx
    ref = None
    for c in (0, 1, 0, 1):
        ctx.set_option("compact_nodes", c)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 11): loop.frame(i)
        ctx.synchronize()
        ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
        img = ctx.download(lib.RAYTRACED)
        if ref is None: ref = img
        print(f"{name} compact_nodes {c}: {ms/k*1e3:.1f} us, identical {np.array_equal(img, ref)}", flush=True)
    loop.close()
