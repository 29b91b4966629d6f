"""One-at-a-time sweep of the queue kernel's runtime knobs around the defaults (all bit-identical)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
loop = HybridFrameLoop(scenes.sponza_proc(), 1920, 1080, 16)
ctx = loop.ctx
base = dict(refill_threshold=16, raygen_early_exit=4, lds_stack_levels=8, raygen_waves_per_block=2, xcd_aware=0)
def t(**o):
    for k, v in {**base, **o}.items(): ctx.set_option(k, v)
    for i in range(3): loop.frame(i)
    ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
    for i in range(3, 15): loop.frame(i)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
    return round(ms / n * 1e3, 1)
print("default", t(), t())
for key, vals in (("refill_threshold", (4, 8, 12, 20, 24, 32)), ("raygen_early_exit", (2, 3, 5, 6)), ("lds_stack_levels", (6, 7, 9, 10, 12)),
                  ("raygen_waves_per_block", (1, 4)), ("xcd_aware", (1,))):
    print(key, [(v, t(**{key: v})) for v in vals], flush=True)
loop.close()
