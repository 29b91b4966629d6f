"""Worker of tests/test_tiling_gloo.py::test_prepared_exchange_replays: tiling.PreparedExchange built once and started
several times on CPU tensors over gloo (the same P2P descriptors the RCCL path replays every frame)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkanhybridrenderer_amd import tiling    # noqa: E402


def main():
    out_path, H, W, halo = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = tiling.StripPlan(rank, world, H, *tiling.strip_bounds(H, world, rank), overlap=halo - 2, halo=halo)
    a = torch.full((H, W, 4), -1.0)
    b = torch.full((H, W, 2), -1.0)
    prepared = tiling.PreparedExchange(dist, [a, b], plan, plan.halo)
    bad = 0
    for frame in range(3):
        for t in (a, b):                     # every rank rewrites its own rows each frame; halo rows keep stale data
            t[plan.row_begin:plan.row_end] = float(100 * frame + rank)
        pending = prepared.start()
        if pending is not None:
            pending.finish()
        for peer, _, (ra, rb) in plan.exchanges(plan.halo):
            for t in (a, b):
                bad += int((t[ra:rb] != float(100 * frame + peer)).any())
        for t in (a, b):
            bad += int((t[plan.row_begin:plan.row_end] != float(100 * frame + rank)).any())
    # C2 with strips of unequal height (H not divisible by the world size): every rank's rows assembled on rank 0
    img = torch.full((H, W, 4), -7.0)
    gather = tiling.StripGather(dist, img, plan)
    for frame in range(2):
        img[plan.row_begin:plan.row_end] = float(10 * frame + rank + 1)
        pending = gather.start()
        if pending is not None:
            pending.finish()
        if rank == 0:
            for r in range(world):
                a, b = tiling.strip_bounds(H, world, r)
                bad += int((gather.full[a:b] != float(10 * frame + r + 1)).any())
    # The same with SCREEN TILES (VHR_TEST_GRID=RxC): rectangles from up to 8 neighbours, column ranges packed into / unpacked from
    # persistent staging tensors by PreparedExchange, rectangles assembled by the gather
    grid = os.environ.get("VHR_TEST_GRID")
    if grid:
        gr, gc = (int(v) for v in grid.split("x"))
        TH, TW = 60, 80
        tp = tiling.make_tile_plan(TW, TH, world, rank, 1, 2, 3, grid=(gr, gc))          # 3 iterations: overlap 6, halos 9 / 10
        ta, tb = torch.full((TH, TW, 4), -1.0), torch.full((TH, TW, 2), -1.0)
        tprep = tiling.PreparedExchange(dist, [ta, tb], tp, (tp.halo_rows, tp.halo_cols))
        x0, x1, y0, y1 = tp.rect
        for frame in range(3):
            for t in (ta, tb):
                t[y0:y1, x0:x1] = float(100 * frame + rank)
            pending = tprep.start()
            if pending is not None:
                pending.finish()
            for peer, _, recv in tp.rect_exchanges(tp.halo_rows, tp.halo_cols):
                if recv:
                    for t in (ta, tb):
                        bad += int((t[recv[2]:recv[3], recv[0]:recv[1]] != float(100 * frame + peer)).any())
            for t in (ta, tb):
                bad += int((t[y0:y1, x0:x1] != float(100 * frame + rank)).any())
        timg = torch.full((TH, TW, 4), -7.0)
        tgather = tiling.StripGather(dist, timg, tp)
        for frame in range(2):
            timg[y0:y1, x0:x1] = float(10 * frame + rank + 1)
            pending = tgather.start()
            if pending is not None:
                pending.finish()
            if rank == 0:
                for r in range(world):
                    a0, a1, b0, b1 = tiling.tile_bounds(TW, TH, gr, gc, r // gc, r % gc)
                    bad += int((tgather.full[b0:b1, a0:a1] != float(10 * frame + r + 1)).any())
    res = torch.tensor([bad], dtype=torch.int64)
    dist.all_reduce(res)
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write(f"{int(res[0])} {len(prepared.ops)}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
