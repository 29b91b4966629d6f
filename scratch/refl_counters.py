"""r4: the mirror-ray launch's in-kernel counters and time, both scenes (bench.reflection_block), plus the frame with the mirror ray.
usage: python scratch/refl_counters.py [scene ...] [option=value ...]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vulkanhybridrenderer_amd import lib as _lib
if os.environ.get("VHR_LIB_VARIANT"): _lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import bench
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
names = [a for a in sys.argv[1:] if "=" not in a] or ["sponza_proc", "bistro_proc"]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for name in names:
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 24, reflections=1)
    for k, v in opts.items():
        loop.ctx.set_option(k, int(v))
    def sync():
        torch.cuda.synchronize(); loop.ctx.synchronize()
    for i in range(4): loop.frame(i)
    sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(4, 20): loop.frame(i)
        sync()
        best = min(best, (time.perf_counter() - t0) / 16 * 1e3)
    blk = bench.reflection_block(loop.ctx, loop, 4, sync)
    loop.ctx.set_kernel_timing(["raygen"]); loop.ctx.kernel_time("raygen", reset=True)
    for i in range(4, 12): loop.frame(i)
    sync()
    ms, n = loop.ctx.kernel_time("raygen")
    import hashlib
    from vulkanhybridrenderer_amd import lib
    loop.frame(5); sync()
    md5 = [hashlib.md5(loop.ctx.download(im).tobytes()).hexdigest()[:10] for im in (lib.REFLECTIONS, lib.RAYTRACED, lib.DENOISED)]
    print(json.dumps({"scene": name, "md5_reflections_raytraced_denoised": md5, "options": opts, "frame_ms_with_mirror_ray": round(best, 4), "raygen_us": round(ms / n * 1e3, 1), "traversal_reflection": blk}), flush=True)
    loop.close()
