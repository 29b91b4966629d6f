import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes, lib
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = 1920, 1080
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 40)
ctx = loop.ctx
def run(label):
    for i in range(4): loop.frame(i)
    torch.cuda.synchronize()
    best = []
    for rep in range(12):
        t0 = time.perf_counter()
        for i in range(4, 36): loop.frame(i)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 32 * 1e3)
    print(f"{label}: median {sorted(best)[len(best) // 2]:.4f} ms/frame (min {min(best):.4f})", flush=True)
ctx.set_kernel_timing(["svgf_atrous"]); run("warm-up"); run("a-trous event pairs, pass timestamps on (bench)")
ctx.set_kernel_timing(False); run("no kernel event pairs, pass timestamps on")
ctx.set_option("pass_timestamps", 2); run("no kernel event pairs, pass timestamps 2 (end-of-frame stamp kernel)")
ctx.set_option("pass_timestamps", 3); run("no kernel event pairs, pass timestamps 3 (event pairs on the dispatch packets)")
ctx.set_option("pass_timestamps", 0); run("no event pairs, no pass timestamps")
ctx.set_kernel_timing(["svgf_atrous"]); run("a-trous event pairs, no pass timestamps")
loop.close()
