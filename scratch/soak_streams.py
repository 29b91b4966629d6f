"""Soak: the default schedule (dead a-trous dispatch on the side stream, mirror ray on its own stream, joined by events without system-scope release) against the
same frames issued strictly in order on one stream, for thousands of frames: Denoised, Reflections and the SVGF history compared every `every` frames without
synchronising in between (a rare ordering or visibility fault would stay in the temporal history).   python scratch/soak_streams.py [frames] [every]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vulkanhybridrenderer_amd import lib, scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop

frames, every = (int(sys.argv[1]) if len(sys.argv) > 1 else 3000), (int(sys.argv[2]) if len(sys.argv) > 2 else 250)
for scene_name, W, H in (("sponza_proc", 1920, 1080), ("bistro_proc", 1280, 720)):
    sc = getattr(scenes, scene_name)()
    loops = {}
    for name, opts in (("default", {}), ("in order", {"svgf_async_unread": 0, "reflection_async": 0})):
        loop = HybridFrameLoop(sc, W, H, 64, reflections=2)
        for k, v in opts.items():
            loop.ctx.set_option(k, v)
        loops[name] = loop
    bad = 0
    pc = loops["default"].pc
    keys = [lib.DENOISED, lib.REFLECTIONS, int(pc["shadow_and_ao_history"]), int(pc["shadow_and_ao_moments_history"])]
    for f in range(frames):
        for loop in loops.values():
            loop.frame(f)
        if (f + 1) % every == 0:
            h = {n: [hashlib.md5(l.ctx.download(k).tobytes()).hexdigest() for k in keys] for n, l in loops.items()}
            same = h["default"] == h["in order"]
            bad += not same
            print(f"{scene_name} {W}x{H} frame {f + 1}: {'identical' if same else 'DIFFERENT ' + str(h)}", flush=True)
    for l in loops.values():
        l.close()
    print(f"{scene_name}: {frames} frames, mismatching checkpoints: {bad}", flush=True)
