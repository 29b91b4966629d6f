"""Multi-process (gloo, CPU) check of the strip / halo logic, plus unit checks of the plan arithmetic."""
import os
import subprocess
import sys

import pytest

from vulkanhybridrenderer_amd import tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_overlap_formula():
    assert tiling.atrous_overlap(5) == 30          # 2 + 4 + 8 + 16: iterations 0..3 feed the published image
    assert tiling.atrous_overlap(4) == 14
    assert tiling.atrous_overlap(2) == 2
    assert tiling.atrous_overlap(1) == 0


def test_plans_partition_the_image():
    for H, N in [(1080, 8), (2160, 8), (1080, 4), (270, 2), (1080, 1)]:
        plans = [tiling.make_plan(H, N, r, 3) for r in range(N)]
        assert plans[0].row_begin == 0 and plans[-1].row_end == H
        assert all(a.row_end == b.row_begin for a, b in zip(plans, plans[1:]))
        if N > 1:
            assert all(p.overlap == 30 and p.halo == 35 for p in plans)
            for p in plans:
                for peer, (sa, sb), (ra, rb) in p.exchanges(p.halo):
                    q = plans[peer]
                    assert q.row_begin <= ra and rb <= q.row_end            # I receive rows the peer owns
                    assert p.row_begin <= sa and sb <= p.row_end            # I send rows I own
                    back = [e for e in q.exchanges(p.halo) if e[0] == p.rank][0]
                    assert back[1] == (ra, rb) and back[2] == (sa, sb)      # and the peer's plan mirrors mine
    assert plans[0].rows == 1080
    with pytest.raises(ValueError):
        tiling.make_plan(256, 8, 0, 4)             # 32-row strips cannot hold a 36-row halo


def test_shrinking_extents():
    # n = 5, E = 30: the rows beyond the strip each iteration still has to produce (the last two feed nothing beyond it)
    assert [tiling.atrous_output_extent(30, 1 << i) for i in range(5)] == [28, 24, 16, 0, 0]
    assert [tiling.atrous_output_extent(14, 1 << i) for i in range(4)] == [12, 8, 0, 0]
    assert tiling.atrous_output_extent(0, 1) == 0


# (world, overlap shortfall, per-iteration shrinking extents, bias on those extents)
@pytest.mark.parametrize("world,shrink,strip_shrink,bias", [(2, 0, 0, 0), (3, 0, 0, 0), (2, 1, 0, 0), (2, 0, 1, 0), (3, 0, 1, 0), (2, 0, 1, -1)])
def test_strips_equal_single_process_gloo(oracle, tmp_path, world, shrink, strip_shrink, bias):
    out = tmp_path / "result.txt"
    port = 29611 + world + 10 * strip_shrink + 20 * (bias != 0) + 40 * shrink
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
               VHR_TEST_SHRINK_OVERLAP=str(shrink), VHR_TEST_STRIP_SHRINK=str(strip_shrink), VHR_TEST_STRIP_SHRINK_BIAS=str(bias))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "tiling_worker.py"), str(out), "64", "120", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad, overlap, halo = out.read_text().split()
    if shrink or bias:      # negative controls: one row less than the derived overlap / extents is already wrong
        assert int(bad) > 0
        return
    assert int(bad) == 0, f"{bad} (rank, frame) pairs differ from the single-process result"
    assert int(overlap) == 30


@pytest.mark.parametrize("world,grid,strip_shrink,shrink,cost", [(4, "2x2", 1, 0, 0), (4, "2x2", 0, 0, 0), (2, "1x2", 1, 0, 0), (4, "2x2", 0, 1, 0), (4, "2x2", 1, 0, 1)])
def test_screen_tiles_equal_single_process_gloo(oracle, tmp_path, world, grid, strip_shrink, shrink, cost):
    """The same check for SCREEN TILES (a grid of rows x cols rectangles, corner neighbours included): every pixel a rank did not
    compute or receive is NaN, each rank's owned rectangle and the frame gathered on rank 0 equal the single-process result; one
    pixel less of overlap (the negative control) is caught.  cost = 1: the grid cut at equal COST (tiling.make_tile_plan(cost=...), round 6): the two columns
    of tiles are cut at different heights, a tile has two neighbours on that side -- placement only, the same images."""
    out = tmp_path / "result.txt"
    port = 29811 + world + 10 * strip_shrink + 40 * shrink + (5 if grid == "1x2" else 0) + 7 * cost
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2", VHR_TEST_GRID=grid, VHR_TEST_COST_MAP="1" if cost else "",
               VHR_TEST_SHRINK_OVERLAP=str(shrink), VHR_TEST_STRIP_SHRINK=str(strip_shrink), VHR_TEST_STRIP_SHRINK_BIAS="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "tiling_worker.py"), str(out)] + (["160", "184", "3"] if cost else ["96", "112", "4"])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)      # (the larger image: room for the cuts to move beside the 33-pixel halos)
    assert r.returncode == 0, r.stderr[-2000:]
    bad, overlap, halo = out.read_text().split()
    if shrink:
        assert int(bad) > 0
        return
    assert int(bad) == 0, f"{bad} (rank, frame) pairs differ from the single-process result"
    assert int(overlap) == 30


@pytest.mark.parametrize("world,grid", [(2, ""), (3, ""), (4, "2x2"), (6, "2x3")])
def test_prepared_exchange_replays(tmp_path, world, grid):
    """The P2P descriptor lists the RCCL path builds once and replays every frame (tiling.PreparedExchange, tiling.StripGather),
    on CPU over gloo, with strips of unequal height (97 rows) -- and, for a grid, with screen tiles as well."""
    out = tmp_path / "prepared.txt"
    port = 29700 + world
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", VHR_TEST_GRID=grid)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "prepared_worker.py"), str(out), "97", "16", "9"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    bad, nops = out.read_text().split()
    assert int(bad) == 0
    assert int(nops) == 4                      # rank 0 (strips): one neighbour x two tensors x (send + recv)


@pytest.mark.parametrize("world", [2, 3])
def test_frame_loop_ordering_with_replayed_exchanges(oracle, tmp_path, world):
    """harness.HybridFrameLoop's per-frame ordering (tiling.StripExchanges: epilogue -> deferred exchange -> next frame's wait)
    over 6 frames with the replayed descriptor lists, the double-buffered moments history and the deferred gather, on CPU:
    every rank's owned rows and every gathered frame equal the single-process result bit for bit."""
    out = tmp_path / "loop.txt"
    port = 29750 + world
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "frame_exchange_worker.py"), str(out), "64", "120", "6"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    bad, gathers_checked, n_prepared, n_gathers, degraded = (int(v) for v in out.read_text().split())
    assert bad == 0, f"{bad} frames differ from the single-process result"
    assert gathers_checked == 5                       # frames 0..4, each checked once its gather had landed
    assert n_prepared == 2 and n_gathers == 1         # one descriptor list per moments buffer, replayed; one gather
    assert degraded == 0


@pytest.mark.parametrize("world,grid,cost_first,skip_move", [(4, "2x2", 0, 0), (4, "2x2", 1, 0), (2, "1x2", 0, 0), (4, "2x2", 0, 1)])
def test_replan_between_frames_gloo(oracle, tmp_path, world, grid, cost_first, skip_move):
    """A re-plan between two frames (tiling.replan_transfers / move_state, round 6): after frame 1 the grid is cut again -- equal pixels -> equal cost, or
    the other way round --, the temporal history, the moments history and the previous normals follow their pixels in one grouped batch, everything
    outside the new rectangle's reach is poisoned with NaN, and frames 2-3 still equal the single-process frames on every rank's NEW rectangle and
    in the frame gathered on rank 0.  skip_move: the negative control -- the new rectangles without the transfer are caught."""
    out = tmp_path / "result.txt"
    port = 29871 + world + 3 * cost_first + (5 if grid == "1x2" else 0) + 11 * skip_move
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2", VHR_TEST_GRID=grid, VHR_TEST_COST_MAP="1" if cost_first else "",
               VHR_TEST_SHRINK_OVERLAP="0", VHR_TEST_STRIP_SHRINK="1", VHR_TEST_STRIP_SHRINK_BIAS="0", VHR_TEST_REPLAN="1", VHR_TEST_REPLAN_SKIP_MOVE="1" if skip_move else "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "tiling_worker.py"), str(out), "160", "184", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad, overlap, halo = out.read_text().split()
    if skip_move:
        assert int(bad) > 0
        return
    assert int(bad) == 0, f"{bad} (rank, frame) pairs differ from the single-process result"


def test_replan_transfers_cover_the_new_rectangles():
    """Every pixel of a rank's new rectangle grown by its halo comes from exactly one old owner (itself included), and what one rank receives from a peer is
    what that peer sends to it -- for 2..8 ranks, equal pixels <-> equal cost, at 1080p."""
    import numpy as np
    W, H = 1920, 1080
    ys, xs = np.mgrid[0:135, 0:240]
    cost = (10 + 500 * np.exp(-((xs - 40) ** 2 + (ys - 100) ** 2) / 400.0)).astype(np.uint32)
    for world in (2, 3, 4, 6, 8):
        a = [tiling.make_tile_plan(W, H, world, r, 3, 4) for r in range(world)]
        b = [tiling.make_tile_plan(W, H, world, r, 3, 4, cost=cost) for r in range(world)]
        for old, new in ((a, b), (b, a)):
            for r in range(world):
                need = tiling._grown(new[r].tile_rect(r), new[r].halo_cols if new[r].grid_cols > 1 else 0, new[r].halo_rows if new[r].grid_rows > 1 else 0, W, H)
                seen = np.zeros((H, W), np.int32)
                own = tiling._intersect(need, old[r].tile_rect(r))
                if own:
                    seen[own[2]:own[3], own[0]:own[1]] += 1
                for peer, send, recv in tiling.replan_transfers(old[r], new[r]):
                    if recv:
                        seen[recv[2]:recv[3], recv[0]:recv[1]] += 1
                    back = [t for t in tiling.replan_transfers(old[peer], new[peer]) if t[0] == r]
                    assert back and back[0][1] == recv and back[0][2] == send
                assert (seen[need[2]:need[3], need[0]:need[1]] == 1).all() and int(seen.sum()) == (need[1] - need[0]) * (need[3] - need[2])
