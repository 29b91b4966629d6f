"""Who sits out the trips of the any-hit queue kernel's node loop (a library built with -DVHR_K1_COUNT_IDLE; statistics instantiation): lane-trips of walkers,
of lanes holding a leaf (what a postponed-leaf step could keep walking) and of lanes without a ray.  VHR_LIB_VARIANT=scratch/_variants/libvhr_k1idle.so"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, lib
lib.LIB_PATH = os.path.abspath(os.environ['VHR_LIB_VARIANT'])
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ["sponza_proc", "bistro_proc"]):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 8)
    ctx = loop.ctx
    for kv in os.environ.get("VHR_OPTS", "").split(","):
        if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for i in range(3): loop.frame(i)
    ctx.set_ray_statistics(True); loop.frame(3); torch.cuda.synchronize(); ctx.synchronize()
    d, t, c = ctx.drain_statistics(), ctx.traversal_statistics(), ctx.traversal_cycles()
    ctx.set_ray_statistics(False)
    w, h, f = d["drain_trips_le4"], d["drain_trips_le8"], d["drain_trips_le16"]
    tot = w + h + f
    print(json.dumps({"scene": name, "node_loop_lane_trips": tot, "walkers": round(w / tot, 4), "holding_a_leaf": round(h / tot, 4), "without_a_ray": round(f / tot, 4),
                      "node_trips_per_wave": round(tot / 64 / c["waves"], 2), "node_visits": t["node_visits"], "leaf_visits": t["leaf_visits"], "tri_tests": t["triangle_tests"],
                      "rays": ctx.ray_statistics()["unique_rays"], "refills_per_wave": round(c["refills"] / c["waves"], 2), "opts": os.environ.get("VHR_OPTS", "")}), flush=True)
    loop.close()
