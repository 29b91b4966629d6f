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
    res = torch.tensor([bad], dtype=torch.int64)
    dist.all_reduce(res)
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write(f"{int(res[0])} {len(prepared.ops)}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
