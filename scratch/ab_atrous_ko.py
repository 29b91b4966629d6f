"""a-trous kernel time per step size for one library (VHR_LIB_VARIANT): the five launches of a 1080p sponza_proc frame, rocprof-free
(dispatch-attached event pairs).  With the knock-out libraries (kernels_svgf.hip -DVHR_ATROUS_KO=n) this gives the attribution table
profiles/r3_atrous_knockouts.txt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"):
    lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
import torch
from vulkanhybridrenderer_amd import scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop
W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
opts = [a.split("=") for a in sys.argv[1:]]
loop = HybridFrameLoop(scenes.sponza_proc(), W, H, 12)
ctx = loop.ctx
for k, v in opts: ctx.set_option(k, int(v))
best = None
for rep in range(3):
    for i in range(3): loop.frame(i)
    ctx.set_kernel_timing(["svgf_atrous", "svgf_temporal"]); ctx.kernel_time("svgf_atrous", reset=True); ctx.kernel_time("svgf_temporal", reset=True)
    for r in range(4):
        for i in range(3, 11): loop.frame(i)
    torch.cuda.synchronize()
    ms, k = ctx.kernel_time("svgf_atrous"); tms, tk = ctx.kernel_time("svgf_temporal"); ctx.set_kernel_timing(False)
    cur = (ms / k * 1e3, tms / tk * 1e3)
    best = cur if best is None or cur[0] < best[0] else best
print(f"{os.environ.get('VHR_LIB_VARIANT', 'default')} {opts}: a-trous {best[0]:.2f} us per launch (mean of the five step sizes), temporal {best[1]:.2f} us", flush=True)
loop.close()
