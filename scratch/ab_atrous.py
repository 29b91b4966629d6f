"""A-trous launch time for option sets in one process (event pairs on every launch, 16 frames each, arms interleaved)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
arms = sys.argv[1:] or [""]
W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 12)
ctx = loop.ctx
ctx.set_option("svgf_async_unread", 0)
defaults = {k: v[0] for k, v in lib.option_table().items()}
parsed = [[a.split("=") for a in arm.split(",") if a] for arm in arms]
touched = {k: defaults[k] for kv in parsed for k, v in kv}
times = {i: [] for i in range(len(arms))}
for rep in range(4):
    for i, kv in enumerate(parsed):
        for k, v in touched.items(): ctx.set_option(k, v)
        for k, v in kv: ctx.set_option(k, int(v))
        for f in range(2): loop.frame(f)
        ctx.set_kernel_timing(["svgf_atrous"]); ctx.kernel_time("svgf_atrous", reset=True)
        for f in range(3, 11): loop.frame(f)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time("svgf_atrous"); ctx.set_kernel_timing(False)
        times[i].append(ms / n * 1e3)
for i, arm in enumerate(arms):
    print(f"[{arm}]: a-trous launch {min(times[i]):.2f} us ({[round(t, 2) for t in times[i]]})", flush=True)
loop.close()
