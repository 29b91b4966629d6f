"""Generates the committed fixtures in tests/golden/ from the CPU oracle (run from the repo root:
python tests/golden/make_golden.py).  kat.json holds the hand-derived known-answer values of SURVEY.md
section 8c (evaluated from the formulas of /root/reference/data/shaders/common.glsl:47-68 and IEEE fp16); the
.npz crops pin the oracle's own outputs so later edits cannot drift silently."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob                      # noqa: E402
from tests.helpers import simple_pfd, synthetic_svgf_inputs   # noqa: E402
from vulkanhybridrenderer_amd import abi, camera, scenes       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

kat = {
    "rng": [
        {"input": 0, "seed": "0xc0a9496a", "states": ["0xd90bc8a8", "0xa3cd8c47", "0x5ae9c9c5", "0x19fa5d8d"],
         "random01": [0.847836, 0.63985515, 0.35512972, 0.10147643]},
        {"input": 1, "seed": "0x27922c9d", "random01": [0.13363731, 0.6163658, 0.99212, 0.3945198]},
        {"input": 12345, "seed": "0x0ddeec13", "random01": [0.8584043, 0.13193703, 0.30500972, 0.32222354]},
        {"input": 8170673, "seed": "0x69c2042e", "random01": [0.006801605, 0.5414963]},
    ],
    "raygen_seed": {"x": 1919, "y": 1079, "H": 1080, "frame": 7, "product": 8170673, "seed": "0x69c2042e"},
    "fp16": {"0.2": "0x3266", "0.8": "0x3a66", "0.7071067811865476": "0x39a8", "2049": "0x6800", "2050": "0x6801", "1.0": "0x3c00"},
}
with open(os.path.join(HERE, "kat.json"), "w") as f:
    json.dump(kat, f, indent=1)

W, H = 48, 40
normals, motion, rt = synthetic_svgf_inputs(W, H, seed=1, motion=(1.25, -0.5))
prev, _, _ = synthetic_svgf_inputs(W, H, seed=1)
rng = np.random.default_rng(2)
integ = np.stack([rng.random((H, W)), rng.random((H, W)), 0.2 * rng.random((H, W)), 0.2 * rng.random((H, W))], -1).astype(np.float16).view(np.uint16)
history = rng.random((H, W, 4)).astype(np.float16).view(np.uint16)
moments = rng.random((H, W, 2)).astype(np.float16).view(np.uint16)
pfd = simple_pfd(W, H)
ti, tm = ob.svgf_temporal(pfd, normals, motion, rt, prev, history, moments)
np.savez_compressed(os.path.join(HERE, "svgf_crops.npz"), W=W, H=H, normals=normals, motion=motion, raytraced=rt, prev_normals=prev,
                    integrated=integ, history=history, moments=moments, atrous_step2=ob.svgf_atrous(pfd, normals, integ, 2),
                    temporal_integrated=ti, temporal_moments=tm)

sc = scenes.tiny_scene()
TW, TH = 64, 40
osc = ob.Scene(sc)
pfd = camera.dolly_frames(sc, TW, TH, 2)[1]
n, m, d = osc.gbuffer(pfd, TW, TH)
sa, refl, mask, rays = osc.raygen(pfd, abi.default_trace_params(), n, d)
np.savez_compressed(os.path.join(HERE, "trace_tiny.npz"), W=TW, H=TH, normals=n, motion=m, depth=d, shadow_ao=sa, reflections=refl,
                    mask=mask, rays=rays, pfd=np.frombuffer(pfd.tobytes(), np.uint8))
# next row f3: composition of the same tiny frame (raw shadow/AO, mirror reflections)
n4, m4, d4, al4 = osc.gbuffer(pfd, TW, TH, with_albedo=True)
comp = ob.composition(pfd, (0, 0, 0), al4, n4, m4, d4, sa, refl)
np.savez_compressed(os.path.join(HERE, "composition_tiny.npz"), albedo=al4, composition=comp)
print("golden fixtures written")
# row f4, screen-space alternatives: ssao.comp / ssao_blur.comp / ssr.comp of the same tiny frame (its G-buffer incl. albedo)
raw = ob.ssao(pfd, n4, d4)
np.savez_compressed(os.path.join(HERE, "screen_space_tiny.npz"), ssao_raw=raw, ssao=ob.ssao_blur(pfd, raw),
                    ssr=ob.ssr(pfd, al4, n4, m4, d4))
print("screen-space fixture written")
