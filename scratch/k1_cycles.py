"""Where the any-hit queue kernel's waves spend their lifetime (the statistics instantiation's cycle counters): set-up / refill / node loop / leaves."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, lib
if os.environ.get('VHR_LIB'): lib.LIB_PATH = os.path.abspath(os.environ['VHR_LIB'])
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ["sponza_proc"]):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for i in range(3): loop.frame(i)
    ctx.set_ray_statistics(True); loop.frame(3); torch.cuda.synchronize(); ctx.synchronize()
    c, t, r = ctx.traversal_cycles(), ctx.traversal_statistics(), ctx.ray_statistics()
    ctx.set_ray_statistics(False)
    tot = c["total"]
    print(json.dumps({"scene": name, **{k: round(c[k] / tot, 3) for k in ("setup", "refill", "nodes", "leaves")}, "other": round(1 - (c["setup"] + c["refill"] + c["nodes"] + c["leaves"]) / tot, 3),
                      "refills_per_wave": round(c["refills"] / c["waves"], 2), "waves": c["waves"], "cycles_per_wave": round(tot / c["waves"]), "rays": r["unique_rays"],
                      "cut_entries_per_wave": round(ctx.drain_statistics()["cut_entries"] / c["waves"], 2), "wave_iterations": t["wave_iterations"], "drain_or_outer_trips": c["drain_iterations"], "node_visits": t["node_visits"], "tri_tests": t["triangle_tests"], "leaf_visits": t["leaf_visits"], "lanes": round(t["active_lane_utilisation"], 3)}), flush=True)
    loop.close()
