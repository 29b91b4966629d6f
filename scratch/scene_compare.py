"""The ray-tracing launches on several scenes, same camera path (r5: the stand-ins, the harder stand-in, and all of them turned off the world axes):
any-hit launch, visits / tests per ray, lanes, tree size; the mirror-ray launch alone; the frame.   python scratch/scene_compare.py [scene ...]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vulkanhybridrenderer_amd import abi, camera, lib, scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
DEFAULT = ["sponza_proc", "sponza_proc_rot", "sponza_hard", "sponza_hard_rot", "bistro_proc", "bistro_proc_rot"]
names = sys.argv[1:]
if "--parity" in names:
    names.remove("--parity")
    from oracle import binding as ob
    from tests.helpers import GpuHybrid, oracle_frames, assert_reflections_identical
    sc = scenes.rotated(scenes.tiny_scene())
    tp = abi.default_trace_params()
    frames, _, _ = oracle_frames(ob, sc, 96, 64, 3, tp)
    for gopts in ({}, {"bvh_presplit": 400}, {"bvh_presplit": 400, "bvh_builder": 0}):
        g = GpuHybrid(sc, 96, 64, trace_params=tp, geometry_options=gopts)
        print("tiny_rot", gopts, g.ctx.bvh_statistics(), "builder", g.ctx.bvh_builder_used() if hasattr(g.ctx, "bvh_builder_used") else "?", flush=True)
        for fr in frames:
            g.frame(fr["pfd"], fr["gbuf"])
            assert np.array_equal(g.ctx.download(lib.RAYTRACED), fr["shadow_ao"]), "tiny_rot: visibility differs from the oracle"
            assert_reflections_identical(g.ctx.download(lib.REFLECTIONS), fr["reflections"])
        g.close()
    print("tiny_rot: visibility and mirror-ray payloads bit-exact against the oracle over 3 frames", flush=True)
for name in (names or DEFAULT):
    name, _, opts = name.partition(":")                 # scene:key=value,key=value  (options read by UpdateGeometry)
    gopts = {k: int(v) for k, v in (kv.split("=") for kv in opts.split(",") if kv)}
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 24, reflections=1, geometry_options=gopts)
    ctx = loop.ctx
    def sync():
        torch.cuda.synchronize(); ctx.synchronize()
    for i in range(4): loop.frame(i)
    sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(4, 20): loop.frame(i)
        sync()
        best = min(best, (time.perf_counter() - t0) / 16 * 1e3)
    blk = bench.reflection_block(ctx, loop, 4, sync)
    ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
    for i in range(4, 12): loop.frame(i)
    sync()
    ms, n = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
    ctx.set_ray_statistics(True); loop.frame(5); sync()
    rs, ts, bv = ctx.ray_statistics(), ctx.traversal_statistics(), ctx.bvh_statistics()
    ctx.set_ray_statistics(False)
    rays = max(1, rs["unique_rays"] - (blk["rays"] if blk else 0))
    print(json.dumps({"scene": name, "options": gopts, "refs": bv["triangles"], "k0_ms": round(ctx.bvh_build_ms(), 1) if hasattr(ctx, "bvh_build_ms") else None, "covered": round(loop.covered_pixels[5] / (1920 * 1080), 3), "any_hit_us": round(ms / n * 1e3, 1), "frame_ms_with_mirror_ray": round(best, 4),
                      "nodes": bv["nodes"], "depth": bv["max_depth"], "mirror_ms_alone": blk and blk["avg_launch_ms"], "mirror_visits_per_ray": blk and blk["node_visits_per_ray"],
                      "mirror_lanes": blk and blk["active_lane_utilisation"]}), flush=True)
    loop.close()
