"""Ray-tracing kernel time for several option sets in ONE process (min of 3 x 8 frames each), bit-identity against the first arm.
usage: python scratch/ab_opts.py "k=v,k=v" "k=v" ...   (an empty string = defaults)"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"):
    lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
arms = sys.argv[1:] or [""]
scene_names = os.environ.get("VHR_SCENES", "sponza_proc,bistro_proc").split(",")
W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
for name in scene_names:
    loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12)
    ctx = loop.ctx
    touched, ref = {}, None
    for arm in arms:
        for k, v in touched.items(): ctx.set_option(k, v)           # back to the defaults recorded below
        kv = [a.split("=") for a in arm.split(",") if a]
        for k, v in kv:
            touched.setdefault(k, {"lds_stack_levels": 8, "raygen_early_exit": 4, "refill_threshold": 16, "raygen_waves_per_block": 2, "raygen_tile_pixels": 64,
                                   "shadow_last": 1, "cut_reach": 1, "raygen_cut": 1, "xcd_aware": 0}.get(k, 0))
            ctx.set_option(k, int(v))
        times = []
        for rep in range(3):
            for i in range(3): loop.frame(i)
            ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
            for i in range(3, 11): loop.frame(i)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
            times.append(ms / 8 * 1e3)
        loop.frame(5); torch.cuda.synchronize()
        md5 = hashlib.md5(ctx.download(lib.RAYTRACED).tobytes()).hexdigest()[:12]
        if ref is None: ref = md5
        print(f"{name} [{arm}]: raygen {min(times):.1f} us ({[round(t, 1) for t in times]}), identical {md5 == ref} md5 {md5}", flush=True)
    loop.close()
