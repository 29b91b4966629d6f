"""r4: lifetimes of the ray-tracing launch's waves (vhr_debug_wave_lifetimes): their distribution, and what pairing two waves in a workgroup costs --
a workgroup's slot is free only when BOTH waves have ended, so the shorter-lived wave's slot idles for the difference.
usage: python scratch/wave_lifetimes.py [scene ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ["sponza_proc", "bistro_proc"]):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12)
    for i in range(8): loop.frame(i)
    torch.cuda.synchronize(); loop.ctx.synchronize()
    t = loop.ctx.wave_lifetimes().astype(np.float64)
    pairs = t.reshape(-1, 2)
    mx, mn = pairs.max(axis=1), pairs.min(axis=1)
    print(json.dumps({"scene": name, "waves": int(t.size), "mean_ticks": round(t.mean()), "median": round(float(np.median(t))), "p90": round(float(np.percentile(t, 90))), "p99": round(float(np.percentile(t, 99))),
                      "max": round(t.max()), "max_over_mean": round(t.max() / t.mean(), 2), "sum_ticks": round(t.sum()),
                      "pairing_idle_share": round(float((mx - mn).sum() / (2 * mx).sum()), 4),
                      "note": "pairing_idle_share = sum over workgroups of (longer - shorter lifetime) / sum of 2 x longer = the share of the occupied wave slots that idle because the workgroup's other wave is still running"}), flush=True)
    loop.close()
