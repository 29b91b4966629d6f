import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 12)
ctx = loop.ctx
for label, tmax, aot in (("normal", 10000.0, 5.0), ("tmax tiny (no traversal)", 0.0100001, 0.0100001)):
    loop.tp["tmax"] = tmax; loop.tp["ao_tmax"] = aot
    ctx.set_trace_params(loop.tp)
    for cut in (1, 0):
        ctx.set_option("raygen_cut", cut)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
        ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
        ts = ctx.traversal_statistics(); ctx.set_ray_statistics(False)
        print(f"{label} cut {cut}: {ms/k*1e3:.1f} us, node visits {ts['node_visits']} wave trips {ts['wave_iterations']}", flush=True)
loop.close()
