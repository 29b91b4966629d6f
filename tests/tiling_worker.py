"""Worker of tests/test_tiling_gloo.py: one rank of an N-strip run of the hot path's schedule on CPU.

The compute bodies are the ORACLE's kernels (this is a test); what is under test is the product's placement logic
(vulkanhybridrenderer_amd/tiling.py): strip bounds, overlap E, history halo Hh, which images are exchanged and
when.  Every row a rank did not compute or receive is poisoned with NaN, so any read of an invalid row shows up
in the owned rows of the result, which must equal the single-process result bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob                                   # noqa: E402
from vulkanhybridrenderer_amd import abi, camera, scenes, tiling    # noqa: E402

NAN16 = np.uint16(0x7e00)


def poison_outside(img, rect):
    """NaN everywhere but [x0, x1) x [y0, y1) (clipped to the image)."""
    x0, x1, y0, y1 = rect
    img[:max(0, y0)] = NAN16
    img[min(img.shape[0], y1):] = NAN16
    img[:, :max(0, x0)] = NAN16
    img[:, min(img.shape[1], x1):] = NAN16


def main():
    out_path, W, H, n_frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    scene = scenes.tiny_scene()
    osc = ob.Scene(scene)
    tp = abi.default_trace_params(reflections=False)
    pfds = camera.dolly_frames(scene, W, H, n_frames)
    gbufs = [osc.gbuffer(p, W, H) for p in pfds]

    def max_motion(ch, extent):
        return max(float(np.nanmax(np.abs(g[1].view(np.float16)[..., ch].astype(np.float32)[g[2] != 0]), initial=0.0)) * extent for g in gbufs[1:])
    grid = os.environ.get("VHR_TEST_GRID", "strips")          # "strips" or "RxC" screen tiles
    grid = "strips" if grid == "strips" else tuple(int(v) for v in grid.split("x"))
    cost = None
    if os.environ.get("VHR_TEST_COST_MAP"):                    # a hot spot low on the left: the grid is cut at equal cost, every column of tiles at its own heights
        ys, xs = np.mgrid[0:(H + 7) // 8, 0:(W + 7) // 8]
        cost = (10 + 500 * np.exp(-((xs - 2) ** 2 + (ys - (H // 8 - 3)) ** 2) / 8.0)).astype(np.uint32)
    plan = tiling.make_tile_plan(W, H, world, rank, int(np.ceil(max_motion(1, H))), int(np.ceil(max_motion(0, W))), grid=grid, cost=cost)
    if cost is not None and plan.grid_rows > 1 and plan.grid_cols > 1:
        assert len(set(plan.row_cuts)) > 1 and plan != tiling.make_tile_plan(W, H, world, rank, int(np.ceil(max_motion(1, H))), int(np.ceil(max_motion(0, W))), grid=grid)
    shrink = int(os.environ.get("VHR_TEST_SHRINK_OVERLAP", "0"))      # negative control: a too-small overlap must be caught
    if shrink:
        import dataclasses
        plan = dataclasses.replace(plan, overlap=plan.overlap - shrink, halo_rows=plan.halo_rows - shrink, halo_cols=plan.halo_cols - shrink)
    x0, x1, y0, y1 = plan.rect
    E = plan.overlap
    comp = plan.computed_rect()                                      # what the SVGF kernels compute
    bx0, bx1, by0, by1 = tiling._grown(plan.rect, plan.halo_cols if plan.grid_cols > 1 else 0, plan.halo_rows if plan.grid_rows > 1 else 0, W, H)   # what the blits copy

    # reference: the whole frame in one piece
    ref_svgf = ob.SVGF(W, H)
    ref = []
    for pfd, g in zip(pfds, gbufs):
        sa, _, _, _ = osc.raygen(pfd, tp, g[0], g[2], want_reflections=False)
        ref.append(ref_svgf.frame(pfd, g[0], g[1], sa))

    A = np.zeros((H, W, 4), np.uint16)
    B = np.zeros((H, W, 4), np.uint16)
    prev_normals = np.zeros((H, W, 4), np.uint16)
    history = np.zeros((H, W, 4), np.uint16)
    moments = np.zeros((H, W, 2), np.uint16)
    worst = 0
    den_t = torch.zeros((H, W, 4), dtype=torch.int16)              # persistent: StripGather's descriptors are built once
    gather = tiling.StripGather(dist, den_t, plan)
    for f, (pfd, g) in enumerate(zip(pfds, gbufs)):
        normals, motion, depth = g
        # Raytrace Pass: the owned rectangle only
        rt, _, _, _ = osc.raygen(pfd, tp, normals, depth, rows=(y0, y1), want_reflections=False)
        poison_outside(rt, plan.rect)
        t = torch.from_numpy(rt)
        tiling.exchange_rows(dist, [t], plan, E)                                  # exchange #1
        # SVGF Denoise Pass on the rectangle grown by E
        x, y = A, B
        integ, mom_new = ob.svgf_temporal(pfd, normals, motion, rt, prev_normals, history, moments)
        poison_outside(integ, comp)
        poison_outside(mom_new, comp)
        x[:] = integ
        moments = mom_new
        for i in range(5):
            out = ob.svgf_atrous(pfd, normals, x, 1 << i)
            if int(os.environ.get("VHR_TEST_STRIP_SHRINK", "0")):      # the product's "strip_shrink_overlap": later iterations compute less
                ext = tiling.atrous_output_extent(E, 1 << i) + int(os.environ.get("VHR_TEST_STRIP_SHRINK_BIAS", "0"))
                poison_outside(out, (x0 - ext if plan.grid_cols > 1 else 0, x1 + ext if plan.grid_cols > 1 else W,
                                     y0 - ext if plan.grid_rows > 1 else 0, y1 + ext if plan.grid_rows > 1 else H))
            else:
                poison_outside(out, comp)
            y[:] = out
            if i == 0:
                history[by0:by1, bx0:bx1] = y[by0:by1, bx0:bx1]
            x, y = y, x
        prev_normals[by0:by1, bx0:bx1] = normals[by0:by1, bx0:bx1]
        denoised = y.copy()
        x, y = y, x
        th, tm = torch.from_numpy(history), torch.from_numpy(moments)
        tiling.exchange_rows(dist, [th, tm], plan, (plan.halo_rows, plan.halo_cols))   # exchange #2
        same = np.array_equal(denoised[y0:y1, x0:x1], ref[f][y0:y1, x0:x1])
        if not same:
            worst += 1
            if os.environ.get("VHR_TEST_VERBOSE"):
                d = (denoised[y0:y1, x0:x1] != ref[f][y0:y1, x0:x1]).any(-1)
                yy, xx = np.nonzero(d)
                print(f"rank {rank} frame {f}: {int(d.sum())} pixels differ in rect {plan.rect}, rows {yy.min() + y0}-{yy.max() + y0} cols {xx.min() + x0}-{xx.max() + x0}", flush=True)
        # C2: every rank's owned rows assembled on rank 0 must be the single-process frame, whatever the other rows held
        den_t.copy_(torch.from_numpy(denoised.view(np.int16)))
        pending = gather.start()
        if pending is not None:
            pending.finish()
        if rank == 0 and world > 1 and not shrink and not int(os.environ.get("VHR_TEST_STRIP_SHRINK_BIAS", "0")):
            if not np.array_equal(gather.full.numpy().view(np.uint16), ref[f]):
                worst += 1
        if os.environ.get("VHR_TEST_REPLAN") and f == int(os.environ["VHR_TEST_REPLAN"]):
            # a re-plan between two frames (tiling.move_state): the grid cut again -- at equal cost if it was cut at equal pixels, and the other way round --,
            # the cross-frame state follows its pixels; everything outside the new rectangle's reach is poisoned before the next frame reads it
            mm = (int(np.ceil(max_motion(1, H))), int(np.ceil(max_motion(0, W))))
            if cost is None:
                ys, xs = np.mgrid[0:(H + 7) // 8, 0:(W + 7) // 8]
                new_cost = (10 + 500 * np.exp(-((xs - 2) ** 2 + (ys - (H // 8 - 3)) ** 2) / 8.0)).astype(np.uint32)
            else:
                new_cost = None
            new_plan = tiling.make_tile_plan(W, H, world, rank, mm[0], mm[1], grid=(plan.grid_rows, plan.grid_cols), cost=new_cost)
            assert world == 1 or new_plan.rect != plan.rect or any(tiling.make_tile_plan(W, H, world, r, mm[0], mm[1], grid=(plan.grid_rows, plan.grid_cols), cost=new_cost).rect !=
                                                                     plan.tile_rect(r) for r in range(world))
            tn = torch.from_numpy(prev_normals)
            if not os.environ.get("VHR_TEST_REPLAN_SKIP_MOVE"):          # (negative control: a re-plan that leaves the state behind must be caught)
                tiling.move_state(dist, [th, tm, tn], plan, new_plan)
            plan = new_plan
            x0, x1, y0, y1 = plan.rect
            comp = plan.computed_rect()
            bx0, bx1, by0, by1 = tiling._grown(plan.rect, plan.halo_cols if plan.grid_cols > 1 else 0, plan.halo_rows if plan.grid_rows > 1 else 0, W, H)
            for img in (history, moments, prev_normals):
                poison_outside(img, (bx0, bx1, by0, by1))
            gather = tiling.StripGather(dist, den_t, plan)
    res = torch.tensor([worst], dtype=torch.int64)
    dist.all_reduce(res)
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write(f"{int(res[0])} {plan.overlap} {plan.halo_rows}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
