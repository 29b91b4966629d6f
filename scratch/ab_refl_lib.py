"""Mirror-ray kernel time (one and two bounces) + the raytraced render path's kernel for ONE library (VHR_LIB_VARIANT=<libvhr_*.so>, default the built one): min of 4 x 8
frames and the md5 of the reflections image -- run once per library to compare two builds.   usage: python scratch/ab_refl_lib.py [scene ...]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"):
    lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
for name in (sys.argv[1:] or ("sponza_proc", "bistro_proc")):
    for refl in (1, 2):
        loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12, reflections=refl)
        ctx = loop.ctx
        ctx.set_option("svgf_async_unread", 0)
        times = []
        for rep in range(4):
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["reflection"]); ctx.kernel_time("reflection", reset=True)
            for i in range(3, 11): loop.frame(i)
            ctx.synchronize()
            b, k2 = ctx.kernel_time("reflection")
            ctx.set_kernel_timing(False)
            times.append(b / max(1, k2) * 1e3)
        loop.frame(5); ctx.synchronize()
        h = hashlib.md5(ctx.download(lib.REFLECTIONS).tobytes()).hexdigest()[:10]
        print(f"{os.path.basename(lib.LIB_PATH)} {name} bounces {refl}: reflection {min(times):.1f} us ({[round(t, 1) for t in times]}) md5 {h}", flush=True)
        loop.close()
