"""What are the mirror-ray pixels beyond 3 fp16 steps of the oracle?  bistro_proc 1080p frame 1, one and two bounces:
how many payloads are bit-identical, how many within 3 steps, and for the rest the largest absolute / relative difference, the pixel's
oracle and GPU values, and what the oracle's closest hit for that pixel's mirror ray looks like when the ray is nudged by one ulp
(a silhouette: a neighbouring triangle at nearly the same t)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import binding as ob                                   # noqa: E402
from tests.helpers import GpuHybrid, f16, ulp16_diff                # noqa: E402
from vulkanhybridrenderer_amd import abi, camera, lib, scenes       # noqa: E402

W, H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
rows = (H // 2 - 135, H // 2 + 135) if "--band" in sys.argv else None
sc = scenes.bistro_proc() if "--sponza" not in sys.argv else scenes.sponza_proc()
osc = ob.Scene(sc)
for bounces in (1, 2):
    tp = abi.default_trace_params(reflections=bounces)
    g = GpuHybrid(sc, W, H, denoise=False, trace_params=tp, gbuffer="standin")
    try:
        pfds = camera.dolly_frames(sc, W, H, 2)
        g.frame(pfds[0])
        g.frame(pfds[1])
        n, d = g.ctx.download(lib.NORMALS), g.ctx.download(lib.DEPTH)
        got = g.ctx.download(lib.REFLECTIONS)
        sa, want, mask, _ = osc.raygen(pfds[1], tp, n, d, rows=rows)
    finally:
        g.close()
    r0, r1 = rows if rows else (0, H)
    a, b = got[r0:r1], want[r0:r1]
    steps = ulp16_diff(a, b).max(-1)
    fa, fb = f16(a), f16(b)
    hit = fb[..., 3] > 0
    print(f"bounces {bounces}: not identical {int((steps != 0).sum())}, max steps {int(steps.max())}")
    print(f"bounces {bounces}: pixels {steps.size}, hits {int(hit.sum())}, identical {(steps == 0).mean():.6f}, <= 3 steps {(steps <= 3).mean():.6f}, "
          f"outliers {int((steps > 3).sum())}")
    out = np.argwhere(steps > 3)
    absd = np.abs(fa - fb).max(-1)
    rel = absd / np.maximum(np.abs(fb).max(-1), 1e-6)
    if len(out):
        print(f"  outliers: max abs {absd[steps > 3].max():.5f}, median abs {np.median(absd[steps > 3]):.5f}, max rel {rel[steps > 3].max():.4f}, median rel {np.median(rel[steps > 3]):.5f}")
        order = np.argsort(-absd[steps > 3])[:12]
        for k in order:
            y, x = out[k]
            print(f"   ({x},{y + r0}) gpu {fa[y, x, :3]} oracle {fb[y, x, :3]} steps {steps[y, x]}")
        # clustering: how many outliers have another outlier within 1 pixel (edges / highlights come in runs)
        ys, xs = out[:, 0], out[:, 1]
        m = np.zeros(steps.shape, bool); m[ys, xs] = True
        nb = np.zeros(steps.shape, int)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy or dx:
                    nb += np.roll(np.roll(m, dy, 0), dx, 1)
        print(f"  outliers with an outlier neighbour: {(nb[m] > 0).mean():.3f}")
        # brightness of outliers vs. all hits (specular highlights are bright)
        lum = fb[..., :3].max(-1)
        print(f"  oracle max channel: outliers median {np.median(lum[m]):.4f}, all hits median {np.median(lum[hit]):.4f}")
