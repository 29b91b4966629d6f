import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, 1920, 1080, 8)
for i in range(4): loop.frame(i)
loop.ctx.set_ray_statistics(True)
for opts in [{}, {"raygen_pregen": 1}, {"refill_threshold": 8}, {"refill_threshold": 32}]:
    for k, v in opts.items(): loop.ctx.set_option(k, v)
    loop.frame(5); torch.cuda.synchronize()
    c = loop.ctx.traversal_cycles(); t = loop.ctx.traversal_statistics()
    tot = c["total"]
    print(opts, {k: round(v / tot, 3) for k, v in c.items() if k in ("setup", "refill", "nodes", "leaves")},
          "refills/wave", round(c["refills"] / max(1, c["waves"]), 1), "waves", c["waves"], "ticks/wave", round(tot / max(1, c["waves"])),
          "util", round(t["active_lane_utilisation"], 3), "trips/wave", round(t["wave_iterations"] / max(1, c["waves"]), 1))
    for k in opts: loop.ctx.set_option(k, {"raygen_pregen": 0, "refill_threshold": 16}[k])
loop.close()
