"""Soak: the scenes on which decision (vi) asks binary64 most often (sponza_hard, sponza_hard_rot: two-triangle walls, slivers, drapery) at 1920x1080, GPU against the
ORACLE -- shadow / AO image and mirror-ray payloads (two bounces) bit for bit -- with the counters of the recomputed pixels.   python scratch/soak_hard_oracle.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from tests.helpers import GpuHybrid, bits_equal_nan_aware
from vulkanhybridrenderer_amd import abi, camera, lib, scenes

ob.build(); ob.lib()
W, H = 1920, 1080
bad = 0
for name in ("sponza_hard", "sponza_hard_rot"):
    sc = getattr(scenes, name)()
    osc = ob.Scene(sc)
    tp = abi.default_trace_params(ao_spp=4, reflections=2)
    g = GpuHybrid(sc, W, H, denoise=False, trace_params=tp)
    g.ctx.set_ray_statistics(True)
    try:
        for pfd in camera.dolly_frames(sc, W, H, 4)[1:]:
            t0 = time.time()
            gb = osc.gbuffer(pfd, W, H)
            sa, refl, mask, rays = osc.raygen(pfd, tp, gb[0], gb[2])
            g.frame(pfd, gb)
            st = g.ctx.binary64_statistics()
            v = np.array_equal(g.ctx.download(lib.RAYTRACED), sa)
            r = bool(bits_equal_nan_aware(g.ctx.download(lib.REFLECTIONS), refl).all())
            bad += (not v) + (not r)
            print(f"{name} frame {int(pfd['frame_index'])}: visibility {'identical' if v else 'DIFFERS'}, reflections {'identical' if r else 'DIFFERS'}; recomputed pixels: "
                  f"any-hit {st['pixels_again']}, mirror {st['mirror_pixels_again']} ({time.time() - t0:.0f} s)", flush=True)
    finally:
        g.close()
print("mismatches:", bad)
