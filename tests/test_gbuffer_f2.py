"""Row f2 of SURVEY.md section 8: the stand-in G-buffer producer honours gbuf.frag's alpha discard (:27-32) and normal
mapping (:35-41).  CPU: the oracle's producer on a purpose-built scene.  GPU: the HIP producer against the oracle."""
import numpy as np
import pytest

from tests import f2_scene
from tests.helpers import f16
from vulkanhybridrenderer_amd import camera, lib

W, H = 128, 80


def test_oracle_discards_and_perturbs(oracle):
    sc = f2_scene.scene()
    osc = oracle.Scene(sc)
    pfd = camera.dolly_frames(sc, W, H, 2)[1]
    n, m, d, al = osc.gbuffer(pfd, W, H, with_albedo=True)
    ids = f16(n)[..., 3][d != 0].astype(int)
    present = set(np.unique(ids).tolist())
    assert f2_scene.PANE not in present                          # albedo.a == 0: every fragment discarded (:30-32)
    assert {f2_scene.WALL, f2_scene.FENCE, f2_scene.FLOOR} <= present
    # the fence's bounding rectangle on screen contains wall pixels (seen through the masked texels) in a checker pattern
    fy, fx = np.nonzero((f16(n)[..., 3] == f2_scene.FENCE) & (d != 0))
    box = f16(n)[fy.min():fy.max() + 1, fx.min():fx.max() + 1, 3]
    frac_wall = float((box == f2_scene.WALL).mean())
    assert 0.3 < frac_wall < 0.7
    # albedo alpha of fence pixels is the opaque texel's (255), never a discarded one
    assert (al[..., 3][(f16(n)[..., 3] == f2_scene.FENCE) & (d != 0)] == 255).all()
    # normal-mapped floor: unit normals that deviate from the geometric +y, varying across the floor
    fl = (f16(n)[..., 3] == f2_scene.FLOOR) & (d != 0)
    nf = f16(n)[fl][:, :3]
    assert np.allclose(np.linalg.norm(nf, axis=1), 1.0, atol=2e-3)
    assert nf[:, 1].min() < 0.95 and nf[:, 1].max() > 0.99 and nf[:, 0].std() > 0.1
    # wall: untouched geometric normal
    assert np.allclose(f16(n)[(f16(n)[..., 3] == f2_scene.WALL) & (d != 0)][:, :3], [0, 0, 1], atol=1e-3)


@pytest.mark.gpu
def test_gpu_producer_matches_oracle(oracle):
    sc = f2_scene.scene()
    osc = oracle.Scene(sc)
    ctx = lib.Context(W, H)
    ctx.upload_scene(sc)
    path = lib.HybridRenderPath(ctx, 0, 0, 2, True, 5, lambda c: c.standin_gbuffer_with_albedo(0))
    path.build()
    try:
        for pfd in camera.dolly_frames(sc, W, H, 2):
            ctx.update_per_frame_ubo(0, pfd)
            ctx.execute(0, 0)
            ctx.synchronize()
        n0, m0, d0, al0 = osc.gbuffer(pfd, W, H, with_albedo=True)
        n, d, al = ctx.download(lib.NORMALS), ctx.download(lib.DEPTH), ctx.download(lib.ALBEDO)
        m = ctx.download(lib.MOTION)
        # bit for bit (round 6): the surface chosen after the discards, the perturbed normals, motion, depth and albedo
        assert np.array_equal(n, n0) and np.array_equal(m, m0) and np.array_equal(d.view(np.uint32), d0.view(np.uint32)) and np.array_equal(al, al0)
        # and the hot path runs on it: shadows of the fence have holes (ray tracing treats the fence as opaque geometry,
        # resource_manager.cpp:633, so this only checks that the pass consumed the perturbed G-buffer without trouble)
        den = f16(ctx.download(lib.DENOISED))
        assert np.isfinite(den).all()
    finally:
        path.destroy()
        ctx.close()
