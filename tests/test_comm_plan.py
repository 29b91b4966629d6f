"""The row arithmetic of the strip decomposition exists twice -- tiling.py (bench.py's host, torch.distributed) and the C
planner of csrc/comm.cpp behind vhr_strip_plan_* (the C++ integrator's host, RCCL inside the library).  They must agree for
every rank: bounds, overlap, halo, neighbour row ranges, shrinking a-trous extents, and the refusal of strips thinner than
the history halo.  No GPU involved."""
import pytest

from vulkanhybridrenderer_amd import lib, tiling


@pytest.mark.parametrize("height", [1080, 2160, 97])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_c_planner_equals_tiling(vhr, height, world):
    for motion, steps in ((0, 5), (3, 5), (7, 4), (1, 2), (0, 1)):
        for rank in range(world):
            try:
                want = tiling.make_plan(height, world, rank, motion, steps)
            except ValueError:
                assert lib.strip_plan(height, world, rank, motion, steps) is None     # thinner than the halo: both refuse
                continue
            got = lib.strip_plan(height, world, rank, motion, steps)
            assert got is not None
            assert (got.rank, got.world, got.height, got.row_begin, got.row_end, got.overlap, got.halo) == \
                   (want.rank, want.world, want.height, want.row_begin, want.row_end, want.overlap, want.halo)
            for n_rows in (want.halo, want.overlap, 1):
                if n_rows:
                    assert lib.strip_plan_exchanges(got, n_rows) == want.exchanges(n_rows)
    L = lib.load()
    for steps in range(0, 8):
        assert L.vhr_atrous_overlap(steps) == tiling.atrous_overlap(steps)
    for overlap in (0, 2, 14, 30, 62):
        for i in range(6):
            assert L.vhr_atrous_output_extent(overlap, 1 << i) == tiling.atrous_output_extent(overlap, 1 << i)


@pytest.mark.gpu
def test_comm_world_size_one_smoke():
    """World size 1 on the GPU box: RCCL loads, the communicator initialises, a frame's exchanges are no-ops that leave the
    images alone, and the strip it sets is the whole image.  (N > 1 has not run on hardware: one GPU per box.)"""
    import numpy as np
    from tests.helpers import GpuHybrid
    from vulkanhybridrenderer_amd import abi, camera, scenes
    W, H = 96, 64
    sc = scenes.tiny_scene()
    g = GpuHybrid(sc, W, H, reflections=False, trace_params=abi.default_trace_params(reflections=False), gbuffer="standin")
    comm = None
    try:
        plan = lib.strip_plan(H, 1, 0, 3)
        comm = lib.Comm(g.ctx, plan, lib.Comm.unique_id())
        pc = g.path.push_constants()
        outs = []
        for use_comm in (False, True):
            for pfd in camera.dolly_frames(sc, W, H, 3):
                if use_comm:
                    comm.finish_frame_exchanges()
                g.frame(pfd)
                if use_comm:
                    comm.start_frame_exchanges(int(pc["shadow_and_ao_history"]), int(pc["shadow_and_ao_moments_history"]), lib.DENOISED, 0, None)
            comm.finish_frame_exchanges()
            g.ctx.synchronize()
            outs.append(g.ctx.download(lib.DENOISED))
        assert np.isfinite(outs[1].view(np.float16).astype(np.float32)).all()
    finally:
        if comm:
            comm.destroy()
        g.close()
