import sys, os
sys.path.insert(0, os.getcwd())
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 16)
ctx = loop.ctx
print(W, H)
for variant, b, xcd in ((4, 64, 1), (4, 64, 0), (3, 8, 1), (4, 8, 1), (4, 64, 1)):
    ctx.set_option("atrous_variant", variant); ctx.set_option("atrous_blocks_per_cu", b); ctx.set_option("atrous_xcd_aware", xcd)
    for i in range(4): loop.frame(i)
    ctx.set_kernel_timing(["svgf_atrous"]); ctx.kernel_time("svgf_atrous", reset=True)
    for i in range(4, 16): loop.frame(i)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_time("svgf_atrous")
    ctx.set_kernel_timing(False)
    print(f"variant {variant} blocks/CU {b} xcd {xcd}: {ms / n * 1e3:.2f} us per launch", flush=True)
loop.close()
