import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import scenes, lib, abi, camera
from tests.helpers import GpuHybrid
W, H = (3840, 2160) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
sc = scenes.bistro_proc()
tp = abi.default_trace_params(ao_spp=16, reflections=2)
pfds = camera.dolly_frames(sc, W, H, 2)
outs = {}
for variant in (1, 0, 1):
    g = GpuHybrid(sc, W, H, shadow=True, ao=True, reflections=True, trace_params=tp, gbuffer="standin")
    g.ctx.set_option("reflection_variant", variant)
    frames = []
    for pfd in pfds:
        g.frame(pfd)
        frames.append((g.ctx.download(lib.REFLECTIONS).copy(), g.ctx.download(lib.RAYTRACED).copy()))
    outs.setdefault(variant, []).append(frames)
    g.close()
a, b, c = outs[1][0], outs[0][0], outs[1][1]
for f in range(2):
    for nm, x, y in (("queue vs per-pixel", a, b), ("queue vs queue again", a, c)):
        d = (x[f][0] != y[f][0]).any(-1)
        print("frame", f, nm, "reflections differ at", int(d.sum()), "pixels", np.argwhere(d)[:6].tolist(), "raytraced differ", int((x[f][1] != y[f][1]).any(-1).sum()))
