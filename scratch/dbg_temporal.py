import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import binding as ob
from vulkanhybridrenderer_amd import lib
from tests.helpers import GpuSvgfHarness, f16, simple_pfd, synthetic_svgf_inputs, ulp16_diff
W, H = 160, 96
motion = (0.0, 0.0)
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
ob.build(); ob.lib()
ref_i, ref_m = ob.svgf_temporal(pfd, normals, motion_img, rt, prev_normals, history, moments)
d = ulp16_diff(integ, ref_i)
print("per-channel mismatch rate", [(d[..., c] != 0).mean() for c in range(4)], "moments", [(ulp16_diff(mom, ref_m)[..., c] != 0).mean() for c in range(2)])
ys, xs, cs = np.nonzero(d)
for y, x, c in list(zip(ys, xs, cs))[:12]:
    print(y, x, c, "gpu", f16(integ)[y, x], "ref", f16(ref_i)[y, x], "rt", f16(rt)[y, x], "hist", f16(history)[y, x], "mom", f16(moments)[y, x])
h.close()
