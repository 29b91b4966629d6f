"""K0: the binned-SAH build on the host ("bvh_builder" 0) against the device's (1, the default): build time, tree size and depth, the ray-tracing kernel's
time on each tree, bit-identity of the image."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ["sponza_proc", "bistro_proc"]):
    scene = getattr(scenes, name)()
    loop = HybridFrameLoop(scene, 1920, 1080, 12)
    ctx = loop.ctx
    ref = None
    for builder in (0, 1, 0, 1, 0, 1):
        ctx.set_option("bvh_builder", builder)
        ctx.upload_scene(scene)
        build_ms, upload_ms = ctx.build_times_ms()
        st = ctx.bvh_statistics()
        times = []
        for rep in range(2):
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            times.append(ms / 8 * 1e3)
        ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
        rs, ts = ctx.ray_statistics(), ctx.traversal_statistics(); ctx.set_ray_statistics(False)
        loop.frame(5); torch.cuda.synchronize()
        md5 = hashlib.md5(ctx.download(lib.RAYTRACED).tobytes()).hexdigest()[:12]
        ref = ref or md5
        n = max(1, rs["unique_rays"])
        print(f"{name} bvh_builder {builder} (used {ctx.bvh_builder_used()}): K0 {build_ms:.1f} ms + upload {upload_ms:.1f} ms, nodes {st['nodes']}, depth {st['max_depth']}, form checks {ctx.bvh_form_checks()}, "
              f"raygen {min(times):.1f} us, node visits/ray {ts['node_visits'] / n:.2f}, tri tests/ray {ts['triangle_tests'] / n:.2f}, overflows {rs['stack_overflows']}, identical {md5 == ref}", flush=True)
    loop.close()
