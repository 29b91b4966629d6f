"""a-trous kernel time per launch for a list of `atrous_variant` values, arms interleaved, every dispatch in recorded order; and whether the
Denoised image and the SVGF storage images after 10 frames are bit-identical to the first arm's.   usage: python scratch/ab_atrous_variant.py 5 6"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
variants = [int(a) for a in sys.argv[1:]] or [5, 6]
times = {v: [] for v in variants}
md5 = {}
for v in variants:                                   # identity: a fresh context per arm, the same 10 frames
    loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 12)
    loop.ctx.set_option("svgf_async_unread", 0); loop.ctx.set_option("atrous_variant", v)
    for i in range(10): loop.frame(i)
    loop.ctx.synchronize()
    md5[v] = hashlib.md5(loop.ctx.download(lib.DENOISED).tobytes()).hexdigest()[:12]
    loop.close()
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 12)
ctx = loop.ctx
ctx.set_option("svgf_async_unread", 0)
for rep in range(5):
    for v in variants:
        ctx.set_option("atrous_variant", v)
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["svgf_atrous"]); ctx.kernel_time("svgf_atrous", reset=True)
        for r in range(3):
            for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("svgf_atrous"); ctx.set_kernel_timing(False)
        times[v].append(ms / k * 1e3)
for v in variants:
    print(f"atrous_variant {v}: {min(times[v]):.2f} us per launch ({[round(t, 2) for t in times[v]]}), Denoised after 10 frames md5 {md5[v]} identical to the first arm {md5[v] == md5[variants[0]]}", flush=True)
loop.close()
