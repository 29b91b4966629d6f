"""Where a wave of the ray-tracing kernel spends its lifetime (in-kernel s_memtime sums of the statistics flavour): set-up, refills, node loop, leaf stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
for name in ([a for a in sys.argv[1:] if "=" not in a] or ("sponza_proc", "bistro_proc")):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for k, v in opts: ctx.set_option(k, int(v))
    for i in range(3): loop.frame(i)
    ctx.set_ray_statistics(True); loop.frame(5); ctx.synchronize()
    cy, ts = ctx.traversal_cycles(), ctx.traversal_statistics()
    T = cy["total"]
    print(f"{name}: waves {cy['waves']}, cycles per wave {T / cy['waves']:.0f} (s_memtime ticks), set-up {cy['setup'] / T:.3f}, refills {cy['refill'] / T:.3f} ({cy['refills'] / cy['waves']:.2f} per wave), "
          f"node loop {cy['nodes'] / T:.3f}, leaf stage {cy['leaves'] / T:.3f}, wave trips {ts['wave_iterations'] / cy['waves']:.1f} per wave", flush=True)
    loop.close()
