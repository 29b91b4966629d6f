"""Ray-tracing kernel time for several option sets in ONE process (min of 3 x 8 frames each), bit-identity against the first arm.
usage: python scratch/ab_opts.py "k=v,k=v" "k=v" ...   (an empty string = defaults)
VHR_KERNEL=reflection: the mirror-ray launch instead (reflections on; VHR_BOUNCES=2 for two), identity of the Reflections image."""
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
    kernel = os.environ.get("VHR_KERNEL", "raygen")
    loop = HybridFrameLoop(getattr(scenes, name)(), W, H, 12, reflections=int(os.environ.get("VHR_BOUNCES", "1")) if kernel == "reflection" else False)
    image = lib.REFLECTIONS if kernel == "reflection" else lib.RAYTRACED
    ctx = loop.ctx
    defaults = {k: v[0] for k, v in lib.option_table().items()}
    touched, ref = {}, None
    parsed = [[a.split("=") for a in arm.split(",") if a] for arm in arms]
    for kv in parsed:
        for k, v in kv: touched[k] = defaults.get(k, 0)
    times = {i: [] for i in range(len(arms))}
    md5s = {}
    reps = int(os.environ.get("VHR_REPS", "4"))
    for rep in range(reps):                       # arms interleaved: the clocks drift over the first seconds of a process
        for i, kv in enumerate(parsed):
            for k, v in touched.items(): ctx.set_option(k, v)
            for k, v in kv: ctx.set_option(k, int(v))
            for f in range(2): loop.frame(f)
            ctx.set_kernel_timing([kernel]); ctx.kernel_time(kernel, reset=True)
            for f in range(3, 11): loop.frame(f)
            torch.cuda.synchronize()
            ms, k = ctx.kernel_time(kernel); ctx.set_kernel_timing(False)
            times[i].append(ms / 8 * 1e3)
            if rep == 0:
                loop.frame(5); torch.cuda.synchronize()
                md5s[i] = hashlib.md5(ctx.download(image).tobytes()).hexdigest()[:12]
    for i, arm in enumerate(arms):
        print(f"{name} [{arm}]: {kernel} {min(times[i]):.1f} us ({[round(t, 1) for t in times[i]]}), identical {md5s[i] == md5s[0]} md5 {md5s[i]}", flush=True)
    loop.close()
