import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
from vulkanhybridrenderer_amd import lib, abi, scenes
from vulkanhybridrenderer_amd.harness import HybridFrameLoop, alias_tensor
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, V = int(sys.argv[1]), int(sys.argv[2]), 4
scene = scenes.sponza_proc()
loop = HybridFrameLoop(scene, W, H, V, reflections=False, device=0, rank=rank, world=world, dist=dist)
ref = HybridFrameLoop(scene, W, H, V, reflections=False, device=0)
print(rank, "plan", loop.plan, "mmr", loop.max_motion_rows, ref.max_motion_rows, flush=True)
pc = loop.pc
for i in range(V):
    loop.frame(i); ref.frame(i); torch.cuda.synchronize()
    y0, y1 = loop.owned_rows()
    for name in (lib.RAYTRACED, lib.DENOISED):
        a = alias_tensor(loop.ctx.transient_info(name)).view(torch.int16).cpu().numpy()
        b = alias_tensor(ref.ctx.transient_info(name)).view(torch.int16).cpu().numpy()
        bad = np.unique(np.argwhere(a[y0:y1] != b[y0:y1])[:, 0]) + y0
        print(rank, "frame", i, name[:12], "bad rows:", (bad.min(), bad.max(), len(bad)) if len(bad) else None, flush=True)
    for key in ("shadow_and_ao_history", "shadow_and_ao_moments_history", "prev_frame_normals_and_object_ids"):
        a = alias_tensor(loop.ctx.storage_info(int(pc[key]))).view(torch.int16).cpu().numpy()
        b = alias_tensor(ref.ctx.storage_info(int(ref.pc[key]))).view(torch.int16).cpu().numpy()
        lo, hi = max(0, y0 - loop.plan.halo), min(H, y1 + loop.plan.halo)
        bad = np.unique(np.argwhere(a[lo:hi] != b[lo:hi])[:, 0]) + lo
        print(rank, "frame", i, key[:22], "bad rows in halo range:", (bad.min(), bad.max(), len(bad)) if len(bad) else None, flush=True)
dist.barrier()
