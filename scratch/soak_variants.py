"""Soak: the work-queue kernels against the literal per-pixel kernels at full size -- every BASELINE stand-in scene, a dozen dolly frames, several ray budgets; the
Raytraced and Reflections images must be identical, and the binary64 counters say how often decision (vi)'s second half was asked.  Not a test (minutes of GPU
time): a bug hunt for the mark-and-recompute paths.   python scratch/soak_variants.py [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import abi, camera, lib, scenes
from tests.helpers import GpuHybrid

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = 1920, 1080
bad = 0
for scene_name in ("sponza_proc", "bistro_proc", "sponza_hard", "sponza_proc_rot", "sponza_hard_rot", "bistro_proc_rot"):
    sc = getattr(scenes, scene_name)()
    for ao_spp, bounces in ((2, 1), (16, 2), (40, 1)):
        tp = abi.default_trace_params(ao_spp=ao_spp, reflections=bounces)
        g = GpuHybrid(sc, W, H, denoise=False, trace_params=tp, gbuffer="standin")
        g.ctx.set_ray_statistics(True)
        again = [0, 0]
        try:
            pfds = camera.dolly_frames(sc, W, H, frames + 1)[1:]
            for pfd in pfds:
                imgs = {}
                for variant in (1, 0):
                    g.ctx.set_option("raygen_variant", variant)
                    g.ctx.set_option("reflection_variant", variant)
                    g.frame(pfd)
                    imgs[variant] = (g.ctx.download(lib.RAYTRACED).copy(), g.ctx.download(lib.REFLECTIONS).copy())
                    if variant == 1:
                        st = g.ctx.binary64_statistics()
                        again[0] += st["pixels_again"]; again[1] += st["mirror_pixels_again"]
                for k, what in enumerate(("Raytraced", "Reflections")):
                    if not np.array_equal(imgs[1][k], imgs[0][k]):
                        bad += 1
                        print(f"MISMATCH {scene_name} ao {ao_spp} bounces {bounces} frame {int(pfd['frame_index'])}: {what} differs in {(imgs[1][k] != imgs[0][k]).any(-1).sum()} pixels", flush=True)
        finally:
            g.close()
        print(f"{scene_name}: ao_spp {ao_spp}, {bounces} bounce(s), {frames} frames: any-hit pixels computed again {again[0]}, mirror pixels {again[1]}", flush=True)
print("mismatches:", bad)
