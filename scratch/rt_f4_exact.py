import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from tests import f2_scene
from tests.test_raytraced_path import _run_gpu, W, H
from vulkanhybridrenderer_amd import camera, scenes
ob.build()
for name, sc in (("f4", f2_scene.scene_f4()),):
    osc = ob.Scene(sc)
    pfds = camera.dolly_frames(sc, W, H, 3)
    for alpha in (False, True):
        got = _run_gpu(sc, pfds, alpha)
        for i, (pfd, (img, presented, stats)) in enumerate(zip(pfds, got)):
            want, rays = osc.raytraced(pfd, W, H, alpha)
            d = np.abs(img.astype(int) - want.astype(int))
            print(name, "alpha", alpha, "frame", i, "channels differing", int((d != 0).sum()), "max", int(d.max()), "pixels > 1 step", int((d.max(-1) > 1).sum()), "rays", stats["unique_rays"], rays, flush=True)
