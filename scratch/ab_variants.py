"""Ray-tracing kernel time per library variant (VHR_LIB_VARIANT) and option set, one process per arm is NOT needed: the library is
picked at import.  usage: VHR_LIB_VARIANT=scratch/_variants/libvhr_x.so python scratch/ab_variants.py [option=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"):
    lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import numpy as np, torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
opts = [a.split("=") for a in sys.argv[1:]]
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12)
    ctx = loop.ctx
    for k, v in opts: ctx.set_option(k, int(v))
    times = []
    for rep in range(3):
        for i in range(3): loop.frame(i)
        ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
        for i in range(3, 11): loop.frame(i)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
        times.append(ms / 8 * 1e3)
    loop.frame(5); torch.cuda.synchronize()
    img = ctx.download(lib.RAYTRACED)
    import hashlib
    print(f"{os.environ.get('VHR_LIB_VARIANT', 'default')} {opts} {name}: raygen {min(times):.1f} us (min of 3 x 8 frames; {[round(t, 1) for t in times]}), image md5 {hashlib.md5(img.tobytes()).hexdigest()[:12]}", flush=True)
    loop.close()
