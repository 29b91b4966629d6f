"""K0 stage times of the device builders (VHR_K0_TRACE=1 prints them)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[2:] or ["sponza_proc", "bistro_proc"]):
    scene = getattr(scenes, name)()
    loop = HybridFrameLoop(scene, 1920, 1080, 12)
    ctx = loop.ctx
    ctx.set_option("bvh_builder", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for rep in range(3):
        ctx.upload_scene(scene)
        print(name, "build, upload ms", ctx.build_times_ms(), flush=True)
    loop.close()
