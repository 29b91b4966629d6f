import sys, hashlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib, camera
import os
if os.environ.get('VHR_LIB_VARIANT'): lib.LIB_PATH = os.path.abspath(os.environ['VHR_LIB_VARIANT'])
for name in ("sponza_proc", "bistro_proc"):
    sc = getattr(scenes, name)()
    W, H = 1920, 1080
    pfds = camera.dolly_frames(sc, W, H, 12)
    ctx = lib.Context(W, H)
    ctx.upload_scene(sc)
    path = lib.RaytracedRenderPath(ctx, use_anyhit_shader=False)
    path.build()
    res = {}
    for rep in range(4):
        for mode in (0, 1):
            ctx.set_option("raygen_cost_order", mode)
            for i in range(3):
                ctx.update_per_frame_ubo(0, pfds[i]); ctx.execute(0, 0)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11):
                ctx.update_per_frame_ubo(0, pfds[i]); ctx.execute(0, 0)
            ctx.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            res.setdefault(mode, []).append(ms / k * 1e3)
            if rep == 0:
                res[("md5", mode)] = hashlib.md5(ctx.download(lib.RAYTRACED_OUTPUT).tobytes()).hexdigest()[:10]
    print(f"{name}: raytraced path kernel, cost order 0: {min(res[0]):.1f} us, 1: {min(res[1]):.1f} us; images identical {res[('md5', 0)] == res[('md5', 1)]}", flush=True)
    path.destroy(); ctx.close()
