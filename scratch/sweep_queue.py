"""Sweep of the queue kernel's two scheduling knobs (refill threshold, early-exit sixteenths) -- VHR_LIB_VARIANT picks the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"):
    lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12)
    ctx = loop.ctx
    for key, values in (("raygen_early_exit", (0, 2, 4, 6, 8, 10)), ("refill_threshold", (4, 8, 16, 24, 32, 48)), ("lds_stack_levels", (4, 6, 8, 12))):
        for v in values:
            ctx.set_option(key, v)
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            print(f"{name} {key}={v}: raygen {ms / 8 * 1e3:.1f} us", flush=True)
        ctx.set_option(key, {"raygen_early_exit": 4, "refill_threshold": 16, "lds_stack_levels": 8}[key])
    loop.close()
