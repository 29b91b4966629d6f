"""Is the ray-tracing kernel bound by its vector-memory path?  Same kernel with 1 / 2 redundant 16-byte loads per node visit
(libraries built with -DVHR_EXTRA_LOADS=n into scratch/_variants, chosen with VHR_LIB_VARIANT), and any option=value pairs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"):
    lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
opts = [a.split("=") for a in sys.argv[1:]]
for name in ("sponza_proc", "bistro_proc"):
    loop = HybridFrameLoop(getattr(scenes, name)(), 1920, 1080, 12)
    ctx = loop.ctx
    for k, v in opts: ctx.set_option(k, int(v))
    for i in range(3): loop.frame(i)
    ctx.set_kernel_timing(["raygen"]); ctx.kernel_time("raygen", reset=True)
    for i in range(3, 11): loop.frame(i)
    torch.cuda.synchronize()
    ms, k = ctx.kernel_time("raygen"); ctx.set_kernel_timing(False)
    print(f"{os.environ.get('VHR_LIB_VARIANT', 'default')} {opts} {name}: raygen {ms / 8 * 1e3:.1f} us per frame", flush=True)
    loop.close()
