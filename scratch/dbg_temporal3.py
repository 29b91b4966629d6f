import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import binding as ob
from vulkanhybridrenderer_amd import lib
from tests.helpers import GpuSvgfHarness, f16, simple_pfd, synthetic_svgf_inputs, ulp16_diff
ob.build(); ob.lib()
W, H = 160, 96
for motion in [(0.0, 0.0), (1.25, -0.5), (-3.5, 2.25)]:
    normals, motion_img, rt = synthetic_svgf_inputs(W, H, seed=7, motion=motion)
    prev_normals, _, _ = synthetic_svgf_inputs(W, H, seed=7)
    rng = np.random.default_rng(5)
    history = rng.random((H, W, 4)).astype(np.float16).view(np.uint16)
    moments = rng.random((H, W, 2)).astype(np.float16).view(np.uint16)
    pfd = simple_pfd(W, H)
    h = None
    def body(ec):
        ec.dispatch(lib.SVGF_SHADER, (W + 7) // 8, (H + 7) // 8, 1, h.push_constants())
    h = GpuSvgfHarness(W, H, body)
    h.ctx.upload(h.images["prev_normals"], prev_normals)
    h.ctx.upload(h.images["history"], history)
    h.ctx.upload(h.images["moments"], moments)
    h.run(pfd, (normals, motion_img, rt))
    integ = h.ctx.download(h.images["a"]); mom = h.ctx.download(h.images["moments"])
    ref_i, ref_m = ob.svgf_temporal(pfd, normals, motion_img, rt, prev_normals, history, moments)
    print(motion, "mismatching channels:", int((ulp16_diff(integ, ref_i) != 0).sum()), int((ulp16_diff(mom, ref_m) != 0).sum()))
    h.close()
