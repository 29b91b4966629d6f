"""How the stand-in G-buffer producer's images differ from the oracle's (bitwise)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import scenes, lib, abi, camera
from oracle import binding as ob
from tests import f2_scene
from tests.helpers import f16
def run(name, sc, W, H, albedo=False):
    osc = ob.Scene(sc)
    ctx = lib.Context(W, H)
    ctx.upload_scene(sc)
    path = lib.HybridRenderPath(ctx, 0, 0, 2, True, 5, (lambda c: c.standin_gbuffer_with_albedo(0)) if albedo else (lambda c: c.standin_gbuffer(0)))
    path.build()
    for pfd in camera.dolly_frames(sc, W, H, 2):
        ctx.update_per_frame_ubo(0, pfd); ctx.execute(0, 0); ctx.synchronize()
    o = osc.gbuffer(pfd, W, H, with_albedo=albedo)
    g = [ctx.download(k) for k in (lib.NORMALS, lib.MOTION, lib.DEPTH)] + ([ctx.download(lib.ALBEDO)] if albedo else [])
    names = ["normals", "motion", "depth", "albedo"]
    for k in range(len(g)):
        a, b = np.asarray(g[k]), np.asarray(o[k])
        d = (a != b)
        if a.dtype == np.float32: d &= ~(np.isnan(a) & np.isnan(b))
        px = d.reshape(H, W, -1).any(-1)
        print(name, names[k], "pixels differing", int(px.sum()), "of", H * W, "first", np.argwhere(px)[:4].tolist())
        if px.any():
            ys, xs = np.nonzero(px)
            for y, x in list(zip(ys, xs))[:4]:
                print("    at", (int(y), int(x)), "gpu", a[y, x].tolist() if a.ndim > 2 else float(a[y, x]).hex(), "oracle", b[y, x].tolist() if b.ndim > 2 else float(b[y, x]).hex())
    path.destroy(); ctx.close()
run("tiny 96x64", scenes.tiny_scene(), 96, 64)
run("f2 scene", f2_scene.scene(), f2_scene.W if hasattr(f2_scene, "W") else 160, f2_scene.H if hasattr(f2_scene, "H") else 96, albedo=True)
run("sponza_proc(0.3) 480x270", scenes.sponza_proc(0.3), 480, 270)
run("sponza_proc 1920x1080", scenes.sponza_proc(), 1920, 1080)
run("bistro_proc(0.2) 480x270", scenes.bistro_proc(0.2), 480, 270, albedo=True)
