import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import lib, scenes, camera
if os.environ.get("VHR_LIB"): lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB"])
sc = scenes.rotated(scenes.sponza_proc(0.3), rot_y=0.6, rot_x=0.25)
Wv, Hv = 300, 170
pfds = camera.dolly_frames(sc, Wv, Hv, 2)
for mode in (0, 1):
    ctx = lib.Context(Wv, Hv)
    try: ctx.set_option("bvh_frame", mode)
    except Exception as e: print("no bvh_frame option"); 
    ctx.upload_scene(sc)
    ctx.set_ray_statistics(True)
    path = lib.RaytracedRenderPath(ctx, use_anyhit_shader=False)
    path.build()
    for alpha in (False, True):
        path.rebuild(alpha)
        for i, pfd in enumerate(pfds):
            ctx.update_per_frame_ubo(0, pfd)
            imgs = {}
            for variant in (0, 1):
                ctx.set_option("raytraced_variant", variant)
                ctx.execute(0, 0); ctx.synchronize()
                imgs[variant] = (ctx.download(lib.RAYTRACED_OUTPUT).copy(), ctx.ray_statistics()["unique_rays"])
            d = (imgs[0][0] != imgs[1][0]).any(-1)
            ys, xs = np.nonzero(d)
            print(f"mode {mode} alpha {alpha} frame {i}: {int(d.sum())} pixels differ, rays {imgs[0][1]} / {imgs[1][1]}", list(zip(xs[:5].tolist(), ys[:5].tolist())),
                  [(imgs[0][0][y, x].tolist(), imgs[1][0][y, x].tolist()) for x, y in list(zip(xs[:3], ys[:3]))], flush=True)
    path.destroy(); ctx.close()
