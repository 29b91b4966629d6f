"""Wall time of vhr_update_geometry (the whole call: uploads, build, self-checks) with either builder."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, lib
for name in ("sponza_proc", "bistro_proc"):
    scene = getattr(scenes, name)()
    c = lib.Context(256, 144)
    for builder in (1, 0, 1, 0):
        c.set_option("bvh_builder", builder)
        t0 = time.perf_counter(); c.upload_scene(scene); c.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        b, u = c.build_times_ms()
        print(f"{name} bvh_builder {builder}: update_geometry {dt:.1f} ms wall (build {b:.1f}, upload {u:.1f}, the rest = self-checks and their copies: {dt - b - u:.1f})", flush=True)
    c.close()
