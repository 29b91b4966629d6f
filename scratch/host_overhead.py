"""Host cost per frame (launch-bound regime) and GPU kernel time of a 1/8 strip."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop

scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, 64, 64, 24)
for i in range(4): loop.frame(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(20):
    for i in range(4, 24): loop.frame(i)
host = (time.perf_counter() - t0) / 400 * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 400 * 1e3
print(f"64x64: host issue {host:.4f} ms/frame, with final sync {total:.4f} ms/frame")
loop.close()

W, H = 1920, 1080
loop = HybridFrameLoop(scene, W, H, 24)
for n in (1, 8):
    plan = tiling.make_plan(H, n, n // 2, loop.max_motion_rows)
    loop.ctx.set_strip(plan.row_begin, plan.row_end, plan.overlap, plan.halo)
    loop.ctx.set_option("trace_overlap", 1 if n > 1 else 0)
    loop.ctx.set_kernel_timing(["raygen", "svgf_temporal", "svgf_atrous", "blit"])
    for k in ("raygen", "svgf_temporal", "svgf_atrous", "blit"): loop.ctx.kernel_time(k, reset=True)
    for i in range(4, 24): loop.frame(i)
    torch.cuda.synchronize()
    kt = {k: loop.ctx.kernel_time(k) for k in ("raygen", "svgf_temporal", "svgf_atrous", "blit")}
    print(n, kt)
    loop.ctx.set_kernel_timing(False)
loop.close()
