"""Which walk disagrees on the far-from-origin mixed-scale scene: GPU flavours against the oracle with and without its BVH."""
import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from vulkanhybridrenderer_amd import scenes, lib, abi, camera
from oracle import binding as ob
from helpers import GpuHybrid
rng = np.random.default_rng(9)
tiny = scenes.tiny_scene()
v = tiny.vertices.copy()
centre = np.float32(float(sys.argv[1]) if len(sys.argv) > 1 else 12345.678)
S = float(sys.argv[3]) if len(sys.argv) > 3 else 1e3
scale = rng.choice(np.array([1.0 / S, 1.0, 1.0, 1.0, S], np.float32), size=(len(v), 1)).astype(np.float32)
if len(sys.argv) > 2 and sys.argv[2] == "noscale": scale = np.float32(1.0)
v["pos"] = v["pos"] * scale + centre
cam = dict(tiny.camera); cam["position"] = tuple(float(c) + float(centre) for c in cam["position"])
far = dataclasses.replace(tiny, name="tiny_far", vertices=v, camera=cam)
W, H = 96, 64
tp = abi.default_trace_params(reflections=False)
osc = ob.Scene(far)
pfd = camera.dolly_frames(far, W, H, 2)[1]
gbuf = osc.gbuffer(pfd, W, H)
sa_bvh = osc.raygen(pfd, tp, gbuf[0], gbuf[2], use_bvh=True, want_reflections=False)[0]
sa_brute = osc.raygen(pfd, tp, gbuf[0], gbuf[2], use_bvh=False, want_reflections=False)[0]
print("oracle bvh vs brute force: differing texels", int((sa_bvh != sa_brute).any(-1).sum()) if sa_bvh.ndim == 3 else int((sa_bvh != sa_brute).sum()))
g = GpuHybrid(far, W, H, denoise=False, trace_params=tp)
for opts in ({}, {"raygen_variant": 0}, {"raygen_cut": 0}, {"compact_nodes": 1}, {"shadow_packet": 1}):
    for k, val in opts.items(): g.ctx.set_option(k, val)
    g.frame(pfd, gbuf)
    got = g.ctx.download(lib.RAYTRACED)
    d1 = (got != sa_bvh); d2 = (got != sa_brute)
    print(opts, "vs oracle bvh:", int(d1.any(-1).sum() if d1.ndim == 3 else d1.sum()), " vs brute force:", int(d2.any(-1).sum() if d2.ndim == 3 else d2.sum()))
    for k in opts: g.ctx.set_option(k, {"raygen_variant": 1, "raygen_cut": 1, "compact_nodes": 0, "shadow_packet": 0}[k])
print("bvh form checks", g.ctx.bvh_form_checks())
g.close()
