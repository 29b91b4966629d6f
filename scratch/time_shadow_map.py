import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from vulkanhybridrenderer_amd import scenes, lib, camera, abi
for name in ("sponza_proc", "bistro_proc"):
    scene = getattr(scenes, name)()
    W, H = 1920, 1080
    ctx = lib.Context(W, H)
    ctx.upload_scene(scene)
    path = lib.HybridRenderPath(ctx, 1, 2, 2, False, 5, None, None)
    path.build()
    ctx.update_per_frame_ubo(0, camera.dolly_frames(scene, W, H, 2)[1])
    for _ in range(2): ctx.standin_shadow_map()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ctx.standin_shadow_map()
    ctx.synchronize()
    sm = ctx.download(lib.SHADOW_MAP)
    print(name, f"shadow map 4096^2: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms, covered {(sm > 0).mean():.3f}")
    path.destroy(); ctx.close()
