import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import lib, scenes, camera
from oracle import binding as ob
sc = scenes.rotated(scenes.sponza_proc(0.3), rot_y=0.6, rot_x=0.25)
Wv, Hv = 300, 170
pfds = camera.dolly_frames(sc, Wv, Hv, 2)
ob.build(); osc = ob.Scene(sc)
out = {}
for mode in (0, 1):
    ctx = lib.Context(Wv, Hv); ctx.set_option("bvh_frame", mode); ctx.upload_scene(sc)
    path = lib.RaytracedRenderPath(ctx, use_anyhit_shader=False); path.build()
    for i, pfd in enumerate(pfds):
        ctx.update_per_frame_ubo(0, pfd)
        for variant in (0, 1):
            ctx.set_option("raytraced_variant", variant); ctx.execute(0, 0); ctx.synchronize()
            out[(mode, i, variant)] = ctx.download(lib.RAYTRACED_OUTPUT).copy()
    path.destroy(); ctx.close()
for i, pfd in enumerate(pfds):
    want, _ = osc.raytraced(pfd, Wv, Hv, False)
    for key in ((0, i, 0), (0, i, 1), (1, i, 0), (1, i, 1)):
        d = (out[key] != want).any(-1)
        big = (np.abs(out[key].astype(int) - want.astype(int)).max(-1) > 1)
        ys, xs = np.nonzero(big)
        print(f"frame {i} mode {key[0]} variant {key[2]}: {int(d.sum())} pixels differ from the oracle, {int(big.sum())} by more than one step", list(zip(xs.tolist(), ys.tolist()))[:6], flush=True)
