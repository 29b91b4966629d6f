"""Row f4: kernel time and ray rate of the raytraced render path's "Raytracing Pass" (primary + shadow rays) at 1080p."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from vulkanhybridrenderer_amd import camera, lib, scenes

W, H = 1920, 1080
for name, sc in (("sponza_proc", scenes.sponza_proc()), ("bistro_proc", scenes.bistro_proc())):
    pfds = camera.dolly_frames(sc, W, H, 12)
    for alpha in (False, True):
        ctx = lib.Context(W, H)
        ctx.upload_scene(sc)
        present = ctx.upload_new_storage_image(W, H, 50)
        path = lib.RaytracedRenderPath(ctx, use_anyhit_shader=alpha, composition_pass=lambda c: c.standin_raytraced_composition(present))
        path.build()
        for pfd in pfds[:4]:
            ctx.update_per_frame_ubo(0, pfd); ctx.execute(0, 0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for pfd in pfds[4:]:
            ctx.update_per_frame_ubo(0, pfd); ctx.execute(0, 0)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 8
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for pfd in pfds[4:]:
            ctx.update_per_frame_ubo(0, pfd); ctx.execute(0, 0)
        kt, n = ctx.kernel_time("raygen")
        ctx.set_kernel_timing(False)
        ctx.set_ray_statistics(True)
        ctx.update_per_frame_ubo(0, pfds[5]); ctx.execute(0, 0); ctx.synchronize()
        st = ctx.ray_statistics()
        print(f"{name} alpha_test={int(alpha)}: frame {dt*1e3:.3f} ms (trace + composition stand-in), trace kernel {kt/n*1e3:.1f} us, "
              f"rays {st['unique_rays']/1e6:.2f} M -> {st['unique_rays']/(kt/n*1e-3)/1e9:.2f} Grays/s, overflows {st['stack_overflows']}", flush=True)
        path.destroy(); ctx.close()
