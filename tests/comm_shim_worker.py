"""One rank of tests/test_comm_shim.py's failure-injection runs: `world` processes on ONE GPU, the library's own exchanges (vhr_comm_*,
csrc/comm.cpp) over tests/rccl_shim (VHR_RCCL_LIBRARY), started WITHOUT a launcher so that nobody ends a rank but the rank itself.
Runs `frames` frames of the tiled hybrid path through harness.HybridFrameLoop(comm="c_abi"); the shim fails one call on one rank
(VHR_RCCL_SHIM_FAIL).  Exit codes: 0 = all frames ran, 3 = an exchange failed and the communicator behaved as vhr_amd.h says (marked
unusable, further starts refused, finish still drains), 4 = the communicator did not come up (on every rank or on none), 5 = anything else."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, frames, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    exchange_raytraced = len(sys.argv) > 5 and sys.argv[5] == "exchange_raytraced"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from vulkanhybridrenderer_amd import lib, scenes
    from vulkanhybridrenderer_amd.harness import CommBringUpError, HybridFrameLoop
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = scenes.tiny_scene()
    try:
        loop = HybridFrameLoop(scene, 160, 144, frames, reflections=1, rank=rank, world=world, dist=dist, comm="c_abi", device=0,
                               trace_overlap=not exchange_raytraced, grid=(2, 2) if world == 4 else None)
    except CommBringUpError as e:
        print(f"rank {rank}: bring-up refused on every rank: {e}", flush=True)
        os._exit(4)
    try:
        for i in range(frames):
            loop.frame(i)
        loop.finish_pending_exchange()
        loop.ctx.synchronize()
    except lib.VhrError as e:
        print(f"rank {rank}: exchange failed: {e}", flush=True)
        comm, pc = loop.comm, loop.pc
        # vhr_amd.h: a failure inside a grouped batch marks the communicator unusable; what was enqueued before it is drained by finish
        ok = True
        try:
            comm.finish_frame_exchanges()
        except lib.VhrError as e2:
            print(f"rank {rank}: finish after the failure raised: {e2}", flush=True)
            ok = False
        try:
            comm.start_frame_exchanges(int(pc["shadow_and_ao_history"]), int(pc["shadow_and_ao_moments_history"]), lib.DENOISED, 0,
                                       loop._gather_buffer.data_ptr() if loop._gather_buffer is not None else None)
            # (a rank whose OWN call never failed -- its peer died -- is not marked broken by the library; its next receive fails instead)
            print(f"rank {rank}: a start after the failure was accepted", flush=True)
        except lib.VhrError as e3:
            print(f"rank {rank}: start after the failure refused: {e3}", flush=True)
        loop.ctx.synchronize()
        os._exit(3 if ok else 5)
    except Exception as e:   # noqa: BLE001
        print(f"rank {rank}: unexpected {e!r}", flush=True)
        os._exit(5)
    print(f"rank {rank}: {frames} frames ran", flush=True)
    loop.close()
    os._exit(0)


if __name__ == "__main__":
    main()
