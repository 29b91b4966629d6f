"""Headless frame loop around the hot path: what Renderer::Render (/root/reference/src/rendering_backend/
renderer.cpp:184-235) does per frame, minus windowing -- fill PerFrameData, run the render graph -- with the
untouched raster stages stood in for: the G-buffer of every frame is produced up front by the stand-in
primary-ray producer and bound as external memory (the route a Vulkan integration would use), so the timed
loop contains only the hot path (Raytrace Pass + SVGF Denoise Pass) and, for N > 1 GPUs, its halo exchanges.

PyTorch is plumbing here: device memory for the precomputed G-buffers, the HIP stream, and torch.distributed
(backend "nccl" == RCCL) for the neighbour exchanges.
"""
import numpy as np

from . import abi, camera, lib, tiling

_TYPESTR = {abi.FORMAT_R16G16B16A16_SFLOAT: ("<f2", 4), abi.FORMAT_R16G16_SFLOAT: ("<f2", 2), abi.FORMAT_D32_SFLOAT: ("<f4", 1)}


class _DeviceArray:
    """Exposes an image of the context as __cuda_array_interface__ so torch can alias it (no copy)."""

    def __init__(self, info):
        typestr, ch = _TYPESTR[info.format]
        shape = (info.height, info.width, ch) if ch > 1 else (info.height, info.width)
        self.__cuda_array_interface__ = dict(shape=shape, typestr=typestr, data=(int(info.device_ptr), False), version=2)


def alias_tensor(info):
    import torch
    return torch.as_tensor(_DeviceArray(info), device="cuda")


# Options by tile size for world > 1 (tools/tile_ceilings.py --sweep, profiles/r5_tile_ceilings.txt): the defaults are tuned on whole 1080p / 4K
# frames; a rank's rectangle of an 8-GPU run is a launch of a few hundred workgroups that does not fill the chip for one round.
# [(computed pixels up to, {option: value})], first match.
TILE_TUNING = []


def tuned_tile_options(computed_pixels, world):
    """The option overrides HybridFrameLoop applies for a rank that computes `computed_pixels` (its rectangle grown by the overlap)."""
    if world <= 1:
        return {}
    for limit, options in TILE_TUNING:
        if computed_pixels <= limit:
            return dict(options)
    return {}


class CommBringUpError(RuntimeError):
    """The library's own RCCL route (vhr_comm_*) did not come up.  Raised on EVERY rank of the job or on none, so the ranks stay in step
    and may all take another route (or all stop)."""


class HybridFrameLoop:
    def __init__(self, scene, width, height, n_frames, shadow=True, ao_spp=2, reflections=False, denoise=True,
                 atrous_steps=5, device=0, rank=0, world=1, dist=None, start_frame_index=0, trace_overlap=True, gather=True,
                 frames_in_flight=1, allow_degraded=False, grid=None, comm="torch", geometry_options=None, tile_cost=None):
        """grid: the screen decomposition for world > 1 -- None = the planner's choice (tiling.choose_grid), "strips" = row strips,
        or (grid_rows, grid_cols).  comm: "torch" = the exchanges through torch.distributed (tiling.StripExchanges), "c_abi" = through
        the library's own RCCL calls (vhr_comm_*, csrc/comm.cpp; the unique id travels over torch.distributed's store).  tile_cost: a map of what the
        frame's rays cost per 8 x 8-pixel cell (Context.tile_cost_map() of a whole-image run, the SAME array on every rank): the grid is cut at equal
        cost instead of equal pixels (vhr_tile_plan_make_weighted; placement only)."""
        import torch
        self.torch = torch
        self.scene, self.W, self.H = scene, width, height
        self.rank, self.world, self.dist = rank, world, dist
        torch.cuda.set_device(device)
        self.stream = torch.cuda.current_stream()
        self.ctx = lib.Context(width, height, device=device, stream=self.stream.cuda_stream)
        for key, value in (geometry_options or {}).items():          # options UpdateGeometry reads ("bvh_presplit", "bvh_builder", ...)
            self.ctx.set_option(key, value)
        self.ctx.upload_scene(scene)
        # frames in flight (vulkan_common.h:9, renderer.cpp:103-146): frame i uses resource index i mod n; the library then
        # issues the front of a frame (G-buffer, Raytrace Pass) on a second stream beside the previous frame's SVGF pass
        self.frames_in_flight = max(1, min(3, int(frames_in_flight)))
        self.ctx.set_option("frames_in_flight", self.frames_in_flight)          # read by the graph build below
        self.tp = abi.default_trace_params(shadow=shadow, ao_spp=ao_spp, reflections=reflections)
        self.ctx.set_trace_params(self.tp)
        # `reflections` = mirror bounces (True / 1 = the reference, 2 = the two-bounce extension).  Second-bounce rays exist
        # only where the first bounce hits; they are left out of the ray counts below (a conservative Mrays/s).
        self.rays_per_pixel = int(shadow) + ao_spp + int(bool(reflections))
        self.reference_rays_per_pixel = 4 * int(shadow) + ao_spp + int(bool(reflections))     # raygen.rgen:38-40 duplicates
        self.pfds = camera.dolly_frames(scene, width, height, n_frames, start_frame_index)
        self.current = 0
        self._aliases = {}
        self._gather_requested, self._allow_degraded = bool(gather), bool(allow_degraded)
        self.exchanges = None             # tiling.StripExchanges, once the strip plan exists (N > 1)
        self.path = lib.HybridRenderPath(self.ctx, shadow_mode=0 if shadow else 2, ambient_occlusion_mode=0 if ao_spp else 2,
                                         reflection_mode=0 if reflections else 2, denoise=denoise, atrous_steps=atrous_steps,
                                         gbuffer_pass=self._gbuffer_pass)
        self.path.build()
        self.denoise = denoise
        self.atrous_steps = atrous_steps
        self._precompute_gbuffers()
        self.tile_cost = tile_cost
        self.plan = tiling.make_tile_plan(width, height, world, rank, self.max_motion_rows, self.max_motion_cols, atrous_steps, grid=grid, cost=tile_cost)
        self.comm_mode, self.comm, self._gather_buffer = comm, None, None
        if world > 1:
            p = self.plan
            if comm == "c_abi":          # vhr_comm_create_tiled sets the tile itself
                self._bring_up_c_abi(p, atrous_steps)
            elif p.grid_cols == 1:
                self.ctx.set_strip(p.row_begin, p.row_end, p.overlap, p.halo_rows)
            else:
                self.ctx.set_tile(p.col_begin, p.col_end, p.row_begin, p.row_end, p.overlap, p.halo_rows, p.halo_cols)
            # trace_overlap: the overlap rows' shadow/AO rays are traced here too (rays are per-pixel independent), which
            # removes exchange #1 from the critical path; only the deferred history exchange remains (tiling.py)
            self.trace_overlap = bool(trace_overlap) and denoise
            self.ctx.set_option("trace_overlap", 1 if self.trace_overlap else 0)
            # the path's own host driver runs the doubling a-trous schedule, so later iterations may compute fewer overlap rows
            self.ctx.set_option("strip_shrink_overlap", 1)
            # the pass epilogues below exchange visibility and SVGF history only: the mirror ray's launch need not be waited for there
            self.ctx.set_option("reflection_async", 2)
            # options by tile size (TILE_TUNING above: the r5 sweep found none worth setting, the hook stays)
            cx0, cx1, cy0, cy1 = p.computed_rect()
            self.tile_options = tuned_tile_options((cx1 - cx0) * (cy1 - cy0), world)
            for key, value in self.tile_options.items():
                self.ctx.set_option(key, value)
            if self.comm is None:
                self.exchanges = tiling.StripExchanges(dist, self.plan, trace_overlap=self.trace_overlap, denoise=denoise, gather=self._gather_requested,
                                                       allow_degraded=self._allow_degraded)
                self.ctx.set_pass_epilogue("Raytrace Pass", self._exchange_raytraced)
                if denoise:
                    self.ctx.set_pass_epilogue("SVGF Denoise Pass", self._exchange_history)
            else:
                if self._gather_requested and denoise and rank == 0:
                    self._gather_buffer = torch.empty((height, width, 4), dtype=torch.float16, device="cuda")
                self.ctx.set_pass_epilogue("Raytrace Pass", self._comm_after_raytrace)
                if denoise:
                    self.ctx.set_pass_epilogue("SVGF Denoise Pass", self._comm_after_svgf)
        self.pc = self.path.push_constants() if denoise else None

    def _bring_up_c_abi(self, p, atrous_steps):
        """Collective-safe bring-up of vhr_comm_*: every rank first probes what can fail locally (RCCL loads and hands out an id, the C
        planner accepts this rank's tile), the ranks exchange those verdicts, and only if all are good does rank 0's id travel and
        ncclCommInitRank run.  After the create call the verdicts are exchanged once more.  Any failure raises CommBringUpError on all
        ranks (with the context closed), never on one rank alone while the others wait in a collective."""
        dist, rank, world = self.dist, self.rank, self.world
        uid, cplan, why = None, None, None
        try:
            lib.comm_use_library_from_environment()          # (VHR_RCCL_LIBRARY: the bench's / the tests' explicit choice of the library to load)
            uid = lib.Comm.unique_id()                       # loads librccl.so, resolves its symbols, ncclGetUniqueId
            cplan = lib.tile_plan(self.W, self.H, world, rank, p.grid_rows, p.grid_cols, self.max_motion_rows, self.max_motion_cols, atrous_steps, cost=self.tile_cost)
            if cplan is None:
                why = "the C planner refuses this rank's tile"
        except Exception as e:   # noqa: BLE001
            why = repr(e)
        verdicts = [None] * world
        dist.all_gather_object(verdicts, why)
        bad = [(r, w) for r, w in enumerate(verdicts) if w]
        if not bad:
            ids = [uid if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                self.comm = lib.Comm(self.ctx, cplan, ids[0])
            except Exception as e:   # noqa: BLE001
                why = repr(e)
            dist.all_gather_object(verdicts, why)
            bad = [(r, w) for r, w in enumerate(verdicts) if w]
        if bad:
            if self.comm:
                self.comm.destroy()
                self.comm = None
            self.path.destroy()
            self.ctx.close()
            raise CommBringUpError("; ".join(f"rank {r}: {w}" for r, w in bad))

    # ---- stand-in for the raster G-buffer stage ----
    def _precompute_gbuffers(self):
        torch = self.torch
        self.gbuffers = []
        covered = []
        max_mv = max_mvx = 0.0
        self._binding = False
        for i, pfd in enumerate(self.pfds):
            self.ctx.update_per_frame_ubo(0, pfd)
            self.ctx.standin_gbuffer(0)
            n = alias_tensor(self.ctx.transient_info(lib.NORMALS)).clone()
            m = alias_tensor(self.ctx.transient_info(lib.MOTION)).clone()
            d = alias_tensor(self.ctx.transient_info(lib.DEPTH)).clone()
            self.gbuffers.append((n, m, d))
            covered.append(int(torch.count_nonzero(d).item()))
            mv = m[..., 1].float()
            mv = torch.nan_to_num(mv, nan=0.0)[d != 0]
            if mv.numel():
                max_mv = max(max_mv, float(mv.abs().max().item()) * self.H)
            mvx = torch.nan_to_num(m[..., 0].float(), nan=0.0)[d != 0]
            if mvx.numel():
                max_mvx = max(max_mvx, float(mvx.abs().max().item()) * self.W)
        self.covered_pixels = covered
        self.max_motion_rows = int(np.ceil(max_mv))
        self.max_motion_cols = int(np.ceil(max_mvx))
        self._binding = True

    def frame_slot(self, i):
        """Which precomputed frame (per-frame data + G-buffer) step i replays.  Beyond the last one the sequence wraps to frame 1,
        not 0: frame 0 carries the reference's zero previous matrices (renderer.cpp:188), i.e. NaN motion vectors, which
        svgf.comp maps to texel (0, 0) -- meaningful on the very first frame only (and row 0 is outside every strip but the first)."""
        n = len(self.pfds)
        return i if i < n else (1 + (i - 1) % (n - 1) if n > 1 else 0)

    def _gbuffer_pass(self, ctx):
        if not self._binding:
            return
        n, m, d = self.gbuffers[self.frame_slot(self.current)]
        ctx.bind_external_image(lib.NORMALS, n.data_ptr())
        ctx.bind_external_image(lib.MOTION, m.data_ptr())
        ctx.bind_external_image(lib.DEPTH, d.data_ptr())

    # ---- multi-GPU halo exchanges (tiling.py) ----
    def _alias(self, info):
        """torch views of context images, cached by device pointer (the moments history alternates between two)."""
        key = int(info.device_ptr)
        t = self._aliases.get(key)
        if t is None:
            t = self._aliases[key] = alias_tensor(info)
        return t

    # the per-frame communication itself lives in tiling.StripExchanges (shared with the CPU gloo test); these are the hooks
    @property
    def gather(self):
        if self.comm:
            return bool(self._gather_requested and self.denoise)
        return self.exchanges.gather if self.exchanges else False

    @property
    def gather_error(self):
        return self.exchanges.gather_error if self.exchanges else None

    @property
    def degraded(self):
        return self.exchanges.degraded if self.exchanges else []

    def finish_pending_exchange(self):
        if self.exchanges:
            self.exchanges.finish_pending()
        if self.comm:
            self.comm.finish_frame_exchanges()

    def gathered_frame(self):
        """Rank 0: the full denoised frame assembled by the last finished gather (a torch tensor), else None."""
        if self.comm:
            return self._gather_buffer
        return self.exchanges.gathered_frame() if self.exchanges else None

    # the same two hooks through the library's own RCCL calls (comm == "c_abi")
    def _comm_after_raytrace(self, ctx):
        self.comm.finish_frame_exchanges()            # the previous frame's exchange #2 and gather land before svgf.comp reads the halo
        if not self.trace_overlap and self.denoise:
            self.comm.exchange_raytraced(lib.RAYTRACED)

    def _comm_after_svgf(self, ctx):
        buf = self._gather_buffer
        self.comm.start_frame_exchanges(int(self.pc["shadow_and_ao_history"]), int(self.pc["shadow_and_ao_moments_history"]),
                                        lib.DENOISED if (self._gather_requested and self.denoise) else None, 0, buf.data_ptr() if buf is not None else None)

    def _exchange_raytraced(self, ctx):           # epilogue of the Raytrace Pass
        self.exchanges.after_raytrace(lambda: self._alias(ctx.transient_info(lib.RAYTRACED)))

    def _exchange_history(self, ctx):             # epilogue of the SVGF Denoise Pass
        den = self._alias(ctx.transient_info(lib.DENOISED))                                   # one instance per frame slot
        hist = self._alias(ctx.storage_info(int(self.pc["shadow_and_ao_history"])))
        mom = self._alias(ctx.storage_info(int(self.pc["shadow_and_ao_moments_history"])))    # the buffer just written
        self.exchanges.after_svgf(den, hist, mom)

    # ---- re-plan between two frames: the grid cut again at equal cost (tiling.make_tile_plan(cost=...), tiling.refine_cost_map) ----
    def replan(self, tile_cost):
        """Cut this rank's rectangle again from `tile_cost` (the SAME map on every rank: the ranks' frame times fed back through
        tiling.refine_cost_map, all-gathered by the caller) and carry the path's cross-frame state -- temporal history, moments history, previous normals
        (hybrid_render_path.cpp:247-262) -- to the new rectangles: one grouped batch of point-to-point transfers (tiling.move_state), each pixel from the rank
        that owned it.  Collective: every rank calls it between the same two frames.  Either route (torch.distributed, or vhr_comm_replan); the grid keeps
        its shape.  Placement only: the frames that follow are the ones the old plan would have produced."""
        if self.world == 1:
            return self.plan
        old = self.plan
        new = tiling.make_tile_plan(self.W, self.H, self.world, self.rank, self.max_motion_rows, self.max_motion_cols, self.atrous_steps,
                                    grid=(old.grid_rows, old.grid_cols), cost=tile_cost)
        if self.comm is not None:                          # the library's own RCCL calls: vhr_comm_replan moves the state and sets the tile
            cplan = lib.tile_plan(self.W, self.H, self.world, self.rank, old.grid_rows, old.grid_cols, self.max_motion_rows, self.max_motion_cols, self.atrous_steps, cost=tile_cost)
            assert (cplan.col_begin, cplan.col_end, cplan.row_begin, cplan.row_end) == new.rect
            self.comm.replan(cplan, *(int(self.pc[k]) for k in ("shadow_and_ao_history", "shadow_and_ao_moments_history", "prev_frame_normals_and_object_ids")))
            self.plan, self.tile_cost = new, tile_cost
            return new
        self.finish_pending_exchange()                     # the last frame's halo exchange and gather have landed
        if self.denoise:
            images = [self._alias(self.ctx.storage_info(int(self.pc[k]))) for k in ("shadow_and_ao_history", "shadow_and_ao_moments_history", "prev_frame_normals_and_object_ids")]
            tiling.move_state(self.dist, images, old, new)
        self.torch.cuda.current_stream().synchronize()
        self.plan, self.tile_cost = new, tile_cost
        if new.grid_cols == 1:
            self.ctx.set_strip(new.row_begin, new.row_end, new.overlap, new.halo_rows)
        else:
            self.ctx.set_tile(new.col_begin, new.col_end, new.row_begin, new.row_end, new.overlap, new.halo_rows, new.halo_cols)
        self.exchanges = tiling.StripExchanges(self.dist, new, trace_overlap=self.trace_overlap, denoise=self.denoise, gather=self._gather_requested,
                                               allow_degraded=self._allow_degraded)
        return new

    # ---- one frame of the hot path ----
    def frame(self, i):
        self.current = i
        idx = i % self.frames_in_flight
        self.ctx.update_per_frame_ubo(idx, self.pfds[self.frame_slot(i)])
        self.ctx.execute(idx, 0)

    # ---- checkpoint / resume (the path's only cross-frame state) ----
    def save_state(self):
        """The SVGF state behind the last frame run (vhr_hybrid_save_state) + the index of the next frame: what a fresh loop over the same
        camera path needs to continue bit-identically (the caller's half of the state -- previous view / projection, frame_index,
        renderer.cpp:187-190,202 -- is a function of the frame index here and travels inside the blob as the last PerFrameData)."""
        self.finish_pending_exchange()
        return {"svgf": self.path.save_state(), "next_frame": self.current + 1}

    def load_state(self, state):
        """Restore a save_state(); returns the index of the frame to run next.  The blob's last PerFrameData must be the one this loop's
        camera path has for that frame (else the histories belong to another sequence)."""
        last = self.path.load_state(state["svgf"])
        nxt = int(state["next_frame"])
        if nxt >= 1 and last.tobytes() != self.pfds[self.frame_slot(nxt - 1)].tobytes():
            raise ValueError("load_state: the checkpoint was taken on another camera path / frame sequence")
        self.current = nxt - 1
        return nxt

    def owned_rows(self):
        return self.plan.row_begin, self.plan.row_end

    def owned_rect(self):
        """(x0, x1, y0, y1) of this rank's screen tile."""
        return self.plan.rect

    def rays_in_frame(self, i, owned_only=True):
        """Unique rays traced by this rank in frame i."""
        if self.world == 1 or not owned_only:
            return self.covered_pixels[self.frame_slot(i)] * self.rays_per_pixel
        x0, x1, y0, y1 = self.plan.rect
        d = self.gbuffers[self.frame_slot(i)][2][y0:y1, x0:x1]
        return int(self.torch.count_nonzero(d).item()) * self.rays_per_pixel

    def close(self):
        self.finish_pending_exchange()
        if self.comm:
            self.ctx.synchronize()
            self.comm.destroy()
            self.comm = None
        self.path.destroy()
        self.ctx.close()
