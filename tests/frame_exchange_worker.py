"""Worker of tests/test_tiling_gloo.py::test_frame_loop_ordering_with_replayed_exchanges: one rank of an N-strip frame loop on
CPU, with the ORDERING of harness.HybridFrameLoop -- Raytrace Pass (owned + overlap rows traced locally) -> its epilogue
(StripExchanges.after_raytrace: the previous frame's deferred exchange #2 and gather land) -> SVGF pass with the shrinking
a-trous extents -> its epilogue (StripExchanges.after_svgf: gather + exchange #2 started, NOT waited for) -> next frame.
The exchanges are the replayed descriptor lists (tiling.PreparedExchange / StripGather) keyed by buffer, the moments history
alternates between two buffers like the product's double-buffered image, every row a rank neither computed nor received is
NaN-poisoned.  Compute bodies are the oracle's kernels (this is a test of the placement and ordering logic)."""
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


def poison_outside(img, a, b):
    img[:max(0, a)] = NAN16
    img[min(img.shape[0], b):] = NAN16


def main():
    out_path, W, H, n_frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    scene = scenes.tiny_scene()
    osc = ob.Scene(scene)
    tp = abi.default_trace_params(reflections=False)
    pfds = camera.dolly_frames(scene, W, H, n_frames)
    gbufs = [osc.gbuffer(p, W, H) for p in pfds]
    mvy = max(float(np.nanmax(np.abs(g[1].view(np.float16)[..., 1].astype(np.float32)[g[2] != 0]), initial=0.0)) * H for g in gbufs[1:])
    plan = tiling.make_plan(H, world, rank, int(np.ceil(mvy)))
    y0, y1, E, Hh = plan.row_begin, plan.row_end, plan.overlap, plan.halo
    c0, c1 = max(0, y0 - E), min(H, y1 + E)               # rows svgf.comp and the first a-trous iteration compute
    b0, b1 = max(0, y0 - Hh), min(H, y1 + Hh)             # rows the blits copy

    ref_svgf = ob.SVGF(W, H)
    ref = []
    for pfd, g in zip(pfds, gbufs):
        sa, _, _, _ = osc.raygen(pfd, tp, g[0], g[2], want_reflections=False)
        ref.append(ref_svgf.frame(pfd, g[0], g[1], sa))

    # persistent images as torch tensors (numpy views share the memory): what the exchanges' descriptors point at
    t_A, t_B = torch.zeros((H, W, 4), dtype=torch.int16), torch.zeros((H, W, 4), dtype=torch.int16)
    t_prev, t_hist = torch.zeros((H, W, 4), dtype=torch.int16), torch.zeros((H, W, 4), dtype=torch.int16)
    t_mom = [torch.zeros((H, W, 2), dtype=torch.int16), torch.zeros((H, W, 2), dtype=torch.int16)]      # double buffer
    t_den = torch.zeros((H, W, 4), dtype=torch.int16)
    A, B, prev_normals, history, den = (t.numpy().view(np.uint16) for t in (t_A, t_B, t_prev, t_hist, t_den))
    moments = [t.numpy().view(np.uint16) for t in t_mom]
    cur = 0                                                # which moments buffer svgf.comp READS this frame
    ex = tiling.StripExchanges(dist, plan, trace_overlap=True, denoise=True, gather=True)
    worst, checked_gathers = 0, 0
    for f, (pfd, g) in enumerate(zip(pfds, gbufs)):
        normals, motion, depth = g
        # Raytrace Pass on the owned rows and the E overlap rows either side ("trace_overlap")
        rt, _, _, _ = osc.raygen(pfd, tp, normals, depth, rows=(c0, c1), want_reflections=False)
        poison_outside(rt, c0, c1)
        ex.after_raytrace()                                # the previous frame's exchange #2 / gather land here
        if f > 0 and rank == 0 and world > 1:              # ... so the frame gathered behind this frame's rays is complete now
            checked_gathers += 1
            if not np.array_equal(ex.gathered_frame().numpy().view(np.uint16), ref[f - 1]):
                worst += 1
        # SVGF Denoise Pass
        x, y = A, B
        integ, mom_new = ob.svgf_temporal(pfd, normals, motion, rt, prev_normals, history, moments[cur])
        poison_outside(integ, c0, c1)
        poison_outside(mom_new, c0, c1)
        x[:] = integ
        moments[1 - cur][:] = mom_new                      # written to the OTHER buffer, then the two flip
        cur = 1 - cur
        for i in range(5):
            out = ob.svgf_atrous(pfd, normals, x, 1 << i)
            ext = tiling.atrous_output_extent(E, 1 << i)   # "strip_shrink_overlap"
            poison_outside(out, y0 - ext, y1 + ext)
            y[:] = out
            if i == 0:
                history[b0:b1] = y[b0:b1]
                poison_outside(history, b0, b1)            # rows beyond the blit are not this rank's to know
            x, y = y, x
        prev_normals[b0:b1] = normals[b0:b1]
        den[:] = y                                         # the image iteration 3 wrote (hybrid_render_path.cpp:322-325)
        poison_outside(den, y0, y1)
        if not np.array_equal(den[y0:y1], ref[f][y0:y1]):
            worst += 1
        ex.after_svgf(t_den, t_hist, t_mom[cur])           # started, not waited for
    ex.finish_pending()
    n_prepared, n_gathers = len(ex._prepared), len(ex._gathers)
    res = torch.tensor([worst, checked_gathers if rank == 0 else 0], dtype=torch.int64)
    dist.all_reduce(res)
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write(f"{int(res[0])} {int(res[1])} {n_prepared} {n_gathers} {len(ex.degraded)}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
