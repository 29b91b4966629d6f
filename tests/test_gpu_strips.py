"""N virtual strips on ONE GPU (N contexts, one thread each, host-side halo copies standing in for RCCL) must
reproduce the single-context result BIT FOR BIT (SURVEY.md section 8e 'Correctness test'), and the committed
golden fixtures must be reproduced by the HIP kernels."""
import os
import threading

import numpy as np
import pytest

from tests.helpers import GpuHybrid, GpuSvgfHarness, f16, simple_pfd, ulp16_diff
from vulkanhybridrenderer_amd import abi, camera, lib, scenes, tiling

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _run_strips(scene, W, H, world, pfds, max_motion_rows, trace_overlap=False, shrink=False):
    tp = abi.default_trace_params(reflections=False)
    plans = [tiling.make_plan(H, world, r, max_motion_rows) for r in range(world)]
    ranks = [GpuHybrid(scene, W, H, reflections=False, trace_params=tp, gbuffer="standin") for _ in range(world)]
    barrier = threading.Barrier(world)
    results = [[] for _ in range(world)]
    errors = []

    def exchange(rank, key_of, n_rows):
        """Synchronous host-side stand-in for tiling.exchange_rows: copy the peer-owned halo rows into my image."""
        g = ranks[rank]
        g.ctx.synchronize()
        barrier.wait()
        for key in key_of(g):
            mine = g.ctx.download(key)
            for peer, _, (ra, rb) in plans[rank].exchanges(n_rows):
                mine[ra:rb] = ranks[peer].ctx.download(key_of(ranks[peer])[key_of(g).index(key)])[ra:rb]
            barrier.wait()              # everyone has read before anyone writes
            g.ctx.upload(key, mine)
            barrier.wait()

    def worker(rank):
        try:
            g, plan = ranks[rank], plans[rank]
            g.ctx.set_strip(plan.row_begin, plan.row_end, plan.overlap, plan.halo)
            pc = g.path.push_constants()
            g.ctx.set_option("trace_overlap", 1 if trace_overlap else 0)
            g.ctx.set_option("strip_shrink_overlap", 1 if shrink else 0)
            if not trace_overlap:
                g.ctx.set_pass_epilogue("Raytrace Pass", lambda c: exchange(rank, lambda h: [lib.RAYTRACED], plan.overlap))
            g.ctx.set_pass_epilogue("SVGF Denoise Pass", lambda c: exchange(
                rank, lambda h: [int(pc["shadow_and_ao_history"]), int(pc["shadow_and_ao_moments_history"])], plan.halo))
            for pfd in pfds:
                g.frame(pfd)
                results[rank].append((g.ctx.download(lib.RAYTRACED)[plan.row_begin:plan.row_end],
                                      g.ctx.download(lib.DENOISED)[plan.row_begin:plan.row_end]))
                barrier.wait()
        except Exception as e:   # noqa: BLE001
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for g in ranks:
        g.close()
    if errors:
        raise errors[0]
    return plans, results


@pytest.mark.parametrize("world,trace_overlap,shrink", [(2, False, False), (3, False, False), (2, True, False), (3, True, False),
                                                        (2, True, True), (3, True, True), (3, False, True)])
def test_virtual_strips_bit_identical(world, trace_overlap, shrink):
    scene = scenes.tiny_scene()
    W, H = 96, 132
    pfds = camera.dolly_frames(scene, W, H, 5)
    single = GpuHybrid(scene, W, H, reflections=False, trace_params=abi.default_trace_params(reflections=False), gbuffer="standin")
    ref = []
    max_mv = 0.0
    try:
        for pfd in pfds:
            single.frame(pfd)
            ref.append((single.ctx.download(lib.RAYTRACED), single.ctx.download(lib.DENOISED)))
            mv = f16(single.ctx.download(lib.MOTION))[..., 1]
            d = single.ctx.download(lib.DEPTH)
            if (d != 0).any():
                max_mv = max(max_mv, float(np.nanmax(np.abs(np.nan_to_num(mv[d != 0])))) * H)
    finally:
        single.close()
    plans, results = _run_strips(scene, W, H, world, pfds, int(np.ceil(max_mv)), trace_overlap, shrink)
    for r, plan in enumerate(plans):
        for f, (rt, den) in enumerate(results[r]):
            assert np.array_equal(rt, ref[f][0][plan.row_begin:plan.row_end]), f"rank {r} frame {f}: raytraced rows differ"
            assert np.array_equal(den, ref[f][1][plan.row_begin:plan.row_end]), f"rank {r} frame {f}: denoised rows differ"


def test_golden_fixtures_on_gpu():
    g = np.load(os.path.join(GOLDEN, "svgf_crops.npz"))
    W, H = int(g["W"]), int(g["H"])
    h = None

    # two separate harness runs keep the two fixtures independent
    def body_t(ec):
        ec.dispatch(lib.SVGF_SHADER, (W + 7) // 8, (H + 7) // 8, 1, h.push_constants())

    def body_a(ec):
        ec.dispatch(lib.ATROUS_SHADER, (W + 7) // 8, (H + 7) // 8, 1, h.push_constants(2))

    h = GpuSvgfHarness(W, H, body_t)
    try:
        h.ctx.upload(h.images["prev_normals"], g["prev_normals"])
        h.ctx.upload(h.images["history"], g["history"])
        h.ctx.upload(h.images["moments"], g["moments"])
        h.run(simple_pfd(W, H), (g["normals"], g["motion"], g["raytraced"]))
        d = ulp16_diff(h.ctx.download(h.images["a"]), g["temporal_integrated"])
        assert d.max() <= 2 and (d == 0).mean() > 0.99
        d = ulp16_diff(h.ctx.download(h.images["moments"]), g["temporal_moments"])
        assert d.max() <= 2 and (d == 0).mean() > 0.99
    finally:
        h.close()
    h = GpuSvgfHarness(W, H, body_a)
    try:
        h.ctx.upload(h.images["a"], g["integrated"])
        h.run(simple_pfd(W, H), (g["normals"], g["motion"], g["raytraced"]))
        d = ulp16_diff(h.ctx.download(h.images["b"]), g["atrous_step2"])
        assert d.max() <= 2 and (d == 0).mean() > 0.99
    finally:
        h.close()
    t = np.load(os.path.join(GOLDEN, "trace_tiny.npz"))
    pfd = np.frombuffer(t["pfd"].tobytes(), abi.per_frame_dtype)[0]
    gh = GpuHybrid(scenes.tiny_scene(), int(t["W"]), int(t["H"]), denoise=False)
    try:
        gh.frame(pfd, (t["normals"], t["motion"], t["depth"]))
        assert np.array_equal(gh.ctx.download(lib.RAYTRACED), t["shadow_ao"])       # visibility: bit-exact vs the committed vector
        a, b = f16(gh.ctx.download(lib.REFLECTIONS)), f16(t["reflections"])
        assert (np.abs(a - b) <= 2.0 ** -9 * np.maximum(np.abs(b), 2.0 ** -14)).all()
    finally:
        gh.close()
