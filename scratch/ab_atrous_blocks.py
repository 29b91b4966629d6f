"""a-trous launch time vs persistent blocks per CU (atrous_blocks_per_cu), whole-frame bench workload."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 16)
ctx = loop.ctx
print(W, H)
for b in (16, 24, 32, 64, 16):
    ctx.set_option("atrous_blocks_per_cu", b)
    for i in range(4): loop.frame(i)
    ctx.set_kernel_timing(["svgf_atrous"]); ctx.kernel_time("svgf_atrous", reset=True)
    for i in range(4, 16): loop.frame(i)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_time("svgf_atrous")
    ctx.set_kernel_timing(False)
    print(f"blocks/CU {b}: {ms / n * 1e3:.2f} us per launch", flush=True)
loop.close()
