import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from vulkanhybridrenderer_amd import abi, camera, lib, scenes
from oracle import binding as ob
from tests.helpers import GpuHybrid, f16
scene = scenes.tiny_scene(); W, H = 96, 64
tp = abi.default_trace_params()
osc = ob.Scene(scene)
pfds = camera.dolly_frames(scene, W, H, 3)
g = GpuHybrid(scene, W, H, denoise=False, trace_params=tp)
for v in (0, 1):
    g.ctx.set_option("raygen_variant", v)
    for i, pfd in enumerate(pfds):
        gb = osc.gbuffer(pfd, W, H)
        sa, refl, mask, rays = osc.raygen(pfd, tp, gb[0], gb[2])
        g.frame(pfd, gb)
        got = g.ctx.download(lib.RAYTRACED)
        bad = np.argwhere(got != sa)
        print("variant", v, "frame", i, "bad", len(bad), bad[:6].tolist(), [ (f16(got)[y,x].tolist(), f16(sa)[y,x].tolist()) for y,x,_ in bad[:3]])
