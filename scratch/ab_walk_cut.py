import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib, camera
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
for name in ("sponza_proc", "bistro_proc"):
    sc = getattr(scenes, name)()
    for bounces in (1, 2):
        loop = HybridFrameLoop(sc, W, H, 12, reflections=bounces)
        ctx = loop.ctx
        ref = None
        for cut in (0, 1):
            ctx.set_option("raygen_cut", cut)
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["reflection"]); ctx.kernel_time("reflection", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("reflection"); ctx.set_kernel_timing(False)
            ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
            rs = ctx.ray_statistics(); ctx.set_ray_statistics(False)
            img = ctx.download(lib.REFLECTIONS)
            if ref is None: ref = img
            print(f"{name} mirror x{bounces} cut {cut}: {ms/k*1e3:.1f} us, identical {np.array_equal(img, ref)}, overflows {rs['stack_overflows']}", flush=True)
        loop.close()
    pfd = camera.dolly_frames(sc, W, H, 2)[1]
    for alpha in (False, True):
        ctx = lib.Context(W, H)
        ctx.upload_scene(sc)
        path = lib.RaytracedRenderPath(ctx, use_anyhit_shader=alpha)
        path.build()
        ctx.update_per_frame_ubo(0, pfd)
        ref = None
        for cut in (0, 1):
            ctx.set_option("raygen_cut", cut)
            for i in range(3): ctx.execute(0, 0)
            ctx.synchronize()
            ts = []
            for i in range(8):
                ctx.execute(0, 0); ctx.synchronize(); ctx.gather_performance_statistics()
                ts.append(ctx.pass_time_ms("Raytracing Pass")[1])
            ctx.set_ray_statistics(True); ctx.execute(0, 0); ctx.synchronize(); rs = ctx.ray_statistics(); ctx.set_ray_statistics(False)
            img = ctx.download(lib.RAYTRACED_OUTPUT)
            if ref is None: ref = img
            print(f"{name} raytraced path alpha {alpha} cut {cut}: {np.median(ts)*1e3:.1f} us, identical {np.array_equal(img, ref)}, overflows {rs['stack_overflows']}", flush=True)
        path.destroy(); ctx.close()
