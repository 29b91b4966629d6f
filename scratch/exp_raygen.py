import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, abi
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
scene = scenes.sponza_proc()
def run(label, **kw):
    loop = HybridFrameLoop(scene, W, H, 8, reflections=False, **{k: v for k, v in kw.items() if k in ("shadow", "ao_spp")})
    ctx = loop.ctx
    if "tmax" in kw:
        tp = loop.tp.copy(); tp["tmax"] = kw["tmax"]; tp["ao_tmax"] = kw["tmax"]; ctx.set_trace_params(tp)
    ctx.set_kernel_timing(True)
    for i in range(2, 8): loop.frame(i)
    ctx.kernel_time("raygen", reset=True)
    for r in range(3):
        for i in range(2, 8): loop.frame(i)
    ms, n = ctx.kernel_time("raygen")
    rays = loop.rays_in_frame(4)
    ctx.set_ray_statistics(True); loop.frame(4); torch.cuda.synchronize()
    ts = ctx.traversal_statistics(); rs = ctx.ray_statistics()
    print(f"{label:28s} {ms/n:.4f} ms  rays {rays/1e6:.2f}M  {rays/(ms/n)/1e6:.2f} Grays/s  nodes/ray {ts['node_visits']/max(1,rs['unique_rays']):.1f} util {ts['active_lane_utilisation']:.2f} iters {ts['wave_iterations']}")
    loop.close()
run("shadow+2ao")
run("shadow only", ao_spp=0)
run("2 ao only", shadow=False)
run("8 ao only", shadow=False, ao_spp=8)
run("shadow+2ao tmax=0.02", tmax=0.02)
