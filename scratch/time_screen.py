import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib, camera, abi
W, H = 1920, 1080
for name in ("sponza_proc", "bistro_proc"):
    scene = getattr(scenes, name)()
    ctx = lib.Context(W, H)
    ctx.upload_scene(scene)
    path = lib.HybridRenderPath(ctx, 2, 1, 1, False, 5, lambda c: c.standin_gbuffer_with_albedo(0), None)
    path.build()
    pfds = camera.dolly_frames(scene, W, H, 12)
    for i, pfd in enumerate(pfds):
        if i == 4:
            ctx.set_kernel_timing(["ssao", "ssao_blur", "ssr"])
            for k in ("ssao", "ssao_blur", "ssr"): ctx.kernel_time(k, reset=True)
        ctx.update_per_frame_ubo(0, pfd); ctx.execute(0, 0)
    ctx.synchronize()
    out = {}
    for k in ("ssao", "ssao_blur", "ssr"):
        ms, n = ctx.kernel_time(k); out[k] = round(ms / n * 1e3, 1)
    refl = ctx.download(lib.SSR).view(np.float16)
    print(name, out, "ssr found", float((refl[..., 3] == 1).mean()))
    path.destroy(); ctx.close()
