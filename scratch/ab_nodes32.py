"""The 32-byte half-precision centre / half-extent nodes (option compact_nodes, two loads per visit) against the 48-byte nodes (three) in the
any-hit queue kernel: time, node visits, wave-level trips, bit-identity.   usage: python scratch/ab_nodes32.py [scene ...] [option=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
names = [a for a in sys.argv[1:] if "=" not in a] or ["sponza_proc", "bistro_proc"]
extra = [a.split("=") for a in sys.argv[1:] if "=" in a]
arms = [dict(compact_nodes=0), dict(compact_nodes=1), dict(compact_nodes=1, lds_stack_levels=6), dict(compact_nodes=1, raygen_early_exit=2), dict(compact_nodes=1, raygen_early_exit=6),
        dict(compact_nodes=1, refill_threshold=24), dict(compact_nodes=1, refill_threshold=8), dict(compact_nodes=0)]
for name in names:
    scene = getattr(scenes, name)()
    loop = HybridFrameLoop(scene, 1920, 1080, 12)
    ctx = loop.ctx
    print(name, "form checks", ctx.bvh_form_checks(), ctx.bvh_statistics(), flush=True)
    ref = None
    defaults = dict(compact_nodes=0, lds_stack_levels=8, raygen_early_exit=4, refill_threshold=16)
    for arm in arms:
        for k, v in defaults.items(): ctx.set_option(k, v)
        for k, v in extra: ctx.set_option(k, int(v))
        for k, v in arm.items(): ctx.set_option(k, v)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
        ctx.set_ray_statistics(True); loop.frame(5); torch.cuda.synchronize()
        rs, ts = ctx.ray_statistics(), ctx.traversal_statistics(); ctx.set_ray_statistics(False)
        loop.frame(5); torch.cuda.synchronize()
        img = ctx.download(lib.RAYTRACED)
        if ref is None: ref = img
        n = max(1, rs["unique_rays"])
        print(f"{name} {arm}: {ms / 8 * 1e3:.1f} us, node visits/ray {ts['node_visits'] / n:.2f}, tri tests/ray {ts['triangle_tests'] / n:.2f}, "
              f"wave trips {ts['wave_iterations']}, overflows {rs['stack_overflows']}, identical {np.array_equal(img, ref)}", flush=True)
    loop.close()
