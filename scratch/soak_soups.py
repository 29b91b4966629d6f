"""Soak: random triangle soups (tests/test_gpu_fuzz.soup -- slivers, interpenetrating and coplanar triangles: every ray grazes something) with seeds the test suite does
not use, GPU against the ORACLE: the shadow / AO image and the mirror ray's payloads (one and two bounces) bit for bit, and how often decision (vi) asked binary64.
Not a test (the oracle's frames take CPU minutes): a bug hunt.     python scratch/soak_soups.py [first_seed] [count]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from tests.test_gpu_fuzz import soup
from tests.helpers import GpuHybrid, oracle_frames, bits_equal_nan_aware
from vulkanhybridrenderer_amd import abi, lib

ob.build(); ob.lib()
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
W, H = 256, 160
bad, asked, t0 = 0, [0, 0], time.time()
for seed in range(first, first + count):
    n_tris = [80, 500, 2500, 7000][seed % 4]
    scene = soup(seed, n_tris, 3 + seed % 7)
    for bounces in (1, 2):
        tp = abi.default_trace_params(ao_spp=2 + 3 * (seed % 3), reflections=bounces)
        frames, _, _ = oracle_frames(ob, scene, W, H, 2, tp, denoise=False)
        g = GpuHybrid(scene, W, H, denoise=False, trace_params=tp)
        g.ctx.set_ray_statistics(True)
        try:
            for i, fr in enumerate(frames):
                g.frame(fr["pfd"], fr["gbuf"])
                st = g.ctx.binary64_statistics()
                asked[0] += st["pixels_again"]; asked[1] += st["mirror_pixels_again"]
                if not np.array_equal(g.ctx.download(lib.RAYTRACED), fr["shadow_ao"]):
                    bad += 1; print(f"MISMATCH seed {seed} bounces {bounces} frame {i}: visibility", flush=True)
                if not bits_equal_nan_aware(g.ctx.download(lib.REFLECTIONS), fr["reflections"]).all():
                    bad += 1; print(f"MISMATCH seed {seed} bounces {bounces} frame {i}: reflections", flush=True)
        finally:
            g.close()
    print(f"seed {seed}: {n_tris} triangles ok ({time.time() - t0:.0f} s; binary64 asked for {asked[0]} any-hit pixels, {asked[1]} mirror pixels so far)", flush=True)
print("mismatches:", bad)
