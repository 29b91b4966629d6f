"""Multi-GPU decomposition of the hot path: the framebuffer is sharded by SCREEN TILES -- a grid of grid_rows x grid_cols
rectangles, row strips being the one-column grid (TilePlan / StripPlan) -- one process per GPU, the scene / BVH / textures /
G-buffer replicated (BASELINE.json north_star; the reference itself is single-GPU).  Below, "rows" of a strip
read "rows and columns" for a tile: the overlap E and the history halo apply on every cut side, and an exchange moves rectangles
(a column range is not contiguous in a row-major image: it travels through a contiguous staging tensor).

Per frame and per rank (strip = rows [y0, y1), E = overlap, Hh = history halo):

  Raytrace Pass      rays for the owned rows (per-pixel independent) and, by default ("trace_overlap"), for the E
                     overlap rows either side as well: re-tracing 30 rows costs less than a neighbour round trip
                     on the critical path (1080p / 8: +44 % of a 70 us kernel vs. an RCCL send/recv + stream sync)
  exchange #1        only with trace_overlap off: E rows of the raw shadow/AO image from each neighbour (RCCL send/recv)
  SVGF Denoise Pass  svgf.comp and every a-trous iteration on rows [y0-E, y1+E): the overlap is RECOMPUTED
                     from valid inputs, so no exchange is needed between iterations
  exchange #2        Hh rows of the temporal history and the moments history from each neighbour (their
                     owners computed them exactly); consumed by NEXT frame's svgf.comp, so it is off the
                     critical path of this frame's output
  gather (C2)        every rank's rows of the denoised image to rank 0 (StripGather), where the frame is assembled for the
                     composition stage; started with exchange #2, finished before the next frame's SVGF pass

Why E: the published image is the output of a-trous iteration n-2 (hybrid_render_path.cpp:322-325 copies the
image the LAST iteration did not write); iteration i reads +-2*2^i rows (svgf_atrous_filter.comp:72-75; the
3x3 variance pre-filter's +-1 is inside that), so the rows a strip needs from svgf.comp's output reach
sum_{i=0}^{n-2} 2*2^i = 2*(2^(n-1)-1) = 30 rows beyond the strip for n = 5.
Why Hh: next frame's svgf.comp runs on [y0-E, y1+E) and reads history / moments / previous normals at the
reprojected position, i.e. up to ceil(max |motion.y| * H) + 2 rows further (svgf.comp:52-60,81-84).

Everything here is placement logic shared by the GPU path (bench.py, RCCL via torch.distributed "nccl") and the
CPU gloo test (tests/test_tiling_gloo.py); results are bit-identical to the single-strip run because the same
kernels run on the same inputs -- only where the rows live differs.
"""
from dataclasses import dataclass


def atrous_overlap(atrous_steps=5):
    """Rows of svgf.comp output a strip needs beyond its own rows (see module docstring)."""
    if atrous_steps < 2:
        return 0
    return 2 * (2 ** (atrous_steps - 1) - 1)


def atrous_output_extent(overlap, step):
    """Rows beyond the strip an a-trous launch with step `step` has to produce under the reference's doubling schedule
    (steps 1, 2, 4, ...; hybrid_render_path.cpp:299-319): the iterations up to and including this one have consumed
    sum 2*2^j = 4*step - 2 rows of validity, and the later ones need exactly what is left.  The product's option
    "strip_shrink_overlap" makes the kernels compute only these rows (n = 5, E = 30: 28, 24, 16, 0, 0)."""
    return max(0, overlap - (4 * step - 2))


@dataclass(frozen=True)
class StripPlan:
    rank: int
    world: int
    height: int
    row_begin: int
    row_end: int
    overlap: int          # E: rows recomputed beyond the strip by the SVGF kernels
    halo: int             # Hh: rows of history / moments fetched from each neighbour

    @property
    def rows(self):
        return self.row_end - self.row_begin

    def neighbours(self):
        """[(peer_rank, rows_i_send (a, b), rows_i_receive (a, b))] for a halo of `n` rows -> see exchanges()."""
        return [r for r in (self.rank - 1, self.rank + 1) if 0 <= r < self.world]

    def exchanges(self, n_rows):
        """Row ranges for an n_rows-deep halo: list of (peer, send (a, b), recv (a, b)), global row indices.
        I send rows I own next to the shared boundary and receive the rows the peer owns next to it."""
        out = []
        if self.rank > 0:
            out.append((self.rank - 1, (self.row_begin, min(self.row_end, self.row_begin + n_rows)),
                        (max(0, self.row_begin - n_rows), self.row_begin)))
        if self.rank < self.world - 1:
            out.append((self.rank + 1, (max(self.row_begin, self.row_end - n_rows), self.row_end),
                        (self.row_end, min(self.height, self.row_end + n_rows))))
        return out


def strip_bounds(height, world, rank):
    """Rows [g*H/N, (g+1)*H/N) -- 135 rows at 1080p / 8, 270 at 4K / 8."""
    return (rank * height) // world, ((rank + 1) * height) // world


def make_plan(height, world, rank, max_motion_rows, atrous_steps=5):
    y0, y1 = strip_bounds(height, world, rank)
    if world == 1:
        return StripPlan(rank, world, height, 0, height, 0, 0)
    overlap = atrous_overlap(atrous_steps)
    halo = overlap + int(max_motion_rows) + 2
    smallest = min(strip_bounds(height, world, r)[1] - strip_bounds(height, world, r)[0] for r in range(world))
    if halo > smallest:
        raise ValueError(f"strips of {smallest} rows are thinner than the {halo}-row history halo "
                         f"(overlap {overlap} + motion {max_motion_rows} + 2): use fewer GPUs or a taller image")
    return StripPlan(rank, world, height, y0, y1, overlap, halo)


def tile_bounds(width, height, grid_rows, grid_cols, tile_row, tile_col):
    """(x0, x1, y0, y1) owned by tile (tile_row, tile_col): columns [c*W/C, (c+1)*W/C), rows [r*H/R, (r+1)*H/R)."""
    return ((tile_col * width) // grid_cols, ((tile_col + 1) * width) // grid_cols, (tile_row * height) // grid_rows, ((tile_row + 1) * height) // grid_rows)


def _grown(rect, dx, dy, width, height):
    x0, x1, y0, y1 = rect
    return (max(0, x0 - dx), min(width, x1 + dx), max(0, y0 - dy), min(height, y1 + dy))


def _intersect(a, b):
    r = (max(a[0], b[0]), min(a[1], b[1]), max(a[2], b[2]), min(a[3], b[3]))
    return r if r[0] < r[1] and r[2] < r[3] else None


def choose_grid(width, height, world, overlap):
    """(grid_rows, grid_cols) with rows * cols == world whose busiest rank computes the fewest pixels: its rectangle grown by the
    overlap on every cut side, clipped to the image; ties go to the squarer grid.  1080p, E = 30: 8 -> 2 x 4 (540 x 570 = +19 % over
    the 480 x 540 owned; 8 row strips compute 1920 x 195 = +44 %)."""
    best = None
    for r in range(1, world + 1):
        if world % r:
            continue
        c = world // r
        if r > height or c > width:
            continue
        worst = 0
        for tr in range(r):
            for tc in range(c):
                g = _grown(tile_bounds(width, height, r, c, tr, tc), overlap if c > 1 else 0, overlap if r > 1 else 0, width, height)
                worst = max(worst, (g[1] - g[0]) * (g[3] - g[2]))
        key = (worst, abs(r - c))
        if best is None or key < best[0]:
            best = (key, (r, c))
    if best is None:
        raise ValueError("no grid fits")
    return best[1]


@dataclass(frozen=True)
class TilePlan:
    rank: int
    world: int
    width: int
    height: int
    grid_rows: int
    grid_cols: int
    col_begin: int
    col_end: int
    row_begin: int
    row_end: int
    overlap: int          # E: pixels recomputed beyond the rectangle by the SVGF kernels, on every cut side
    halo_rows: int        # margin of history / moments fetched from the neighbours (E on an axis that is not cut)
    halo_cols: int
    col_cuts: tuple = ()  # the grid's cut lines: tile (r, c) owns columns [col_cuts[c], col_cuts[c + 1]) and rows [row_cuts[c][r], row_cuts[c][r + 1]):
    row_cuts: tuple = ()  # every column of tiles has its own row cuts (equal pixels: the same in all; () = equal pixels, plans made by hand)

    def tile_rect(self, rank):
        """(x0, x1, y0, y1) owned by `rank`."""
        tr, tc = rank // self.grid_cols, rank % self.grid_cols
        if self.col_cuts and self.row_cuts:
            return (self.col_cuts[tc], self.col_cuts[tc + 1], self.row_cuts[tc][tr], self.row_cuts[tc][tr + 1])
        return tile_bounds(self.width, self.height, self.grid_rows, self.grid_cols, tr, tc)

    @property
    def halo(self):
        return self.halo_rows

    @property
    def rect(self):
        return (self.col_begin, self.col_end, self.row_begin, self.row_end)

    def computed_rect(self, extend=None):
        """The rectangle the SVGF kernels compute: the owned one grown by `extend` (default: the overlap) on every cut side."""
        e = self.overlap if extend is None else extend
        return _grown(self.rect, e if self.grid_cols > 1 else 0, e if self.grid_rows > 1 else 0, self.width, self.height)

    def rect_exchanges(self, halo_rows, halo_cols):
        """[(peer, send rect or None, recv rect or None)]: what a margin of (halo_rows, halo_cols) pixels takes from each peer = the
        peer's owned pixels inside my grown rectangle, and gives = my owned pixels inside the peer's grown rectangle."""
        if self.world <= 1 or (halo_rows == 0 and halo_cols == 0):
            return []
        dx, dy = (halo_cols if self.grid_cols > 1 else 0), (halo_rows if self.grid_rows > 1 else 0)
        mine = self.rect
        my_need = _grown(mine, dx, dy, self.width, self.height)
        out = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            theirs = self.tile_rect(peer)
            recv, send = _intersect(my_need, theirs), _intersect(_grown(theirs, dx, dy, self.width, self.height), mine)
            if recv or send:
                out.append((peer, send, recv))
        return out


def balanced_cuts(marginal, cell, extent, n, min_px):
    """n + 1 cut lines of [0, extent) at equal cost (csrc/comm.cpp balanced_cuts, line for line): cut j = the first cell boundary at which the running sum of
    `marginal` (cost per cell-pixel-wide slab) reaches j / n of the total, then moved so that no tile is thinner than min_px; no map = equal pixels."""
    total = int(sum(int(v) for v in marginal)) if marginal is not None else 0
    cuts = [0] * (n + 1)
    cuts[n] = extent
    weighted = total != 0 and len(marginal) != 0
    if not weighted:
        for j in range(1, n):
            cuts[j] = j * extent // n
    else:
        run, b = 0, 0
        for j in range(1, n):
            while b < len(marginal) and run * n < total * j:
                run += int(marginal[b])
                b += 1
            cuts[j] = min(extent, b * cell)
    if n > 1 and min_px * n > extent:
        return None
    if weighted:
        for j in range(1, n):
            cuts[j] = max(cuts[j], cuts[j - 1] + max(1, min_px))
        for j in range(n - 1, 0, -1):
            cuts[j] = min(cuts[j], cuts[j + 1] - max(1, min_px))
    for j in range(1, n + 1):
        if cuts[j] <= cuts[j - 1] or cuts[j] - cuts[j - 1] < min_px:
            return None
    return tuple(cuts)


def make_tile_plan(width, height, world, rank, max_motion_rows=0, max_motion_cols=0, atrous_steps=5, grid=None, cost=None, cost_cell=8):
    """grid: None = choose_grid, "strips" = row strips, or (grid_rows, grid_cols).  cost: a 2-D array, cost[cy, cx] = what the cost_cell x cost_cell pixel
    block at (cx, cy) * cost_cell costs to trace (the any-hit queue kernel's wave lifetimes, say): the grid is cut at equal cost instead of equal pixels
    (vhr_tile_plan_make_weighted; placement only -- the images do not change)."""
    overlap = atrous_overlap(atrous_steps) if world > 1 else 0
    if grid is None:
        grid = choose_grid(width, height, world, overlap)
    elif grid == "strips":
        grid = (world, 1)
    gr, gc = grid
    if gr * gc != world or gr > height or gc > width:
        raise ValueError(f"a {gr} x {gc} grid does not hold {world} ranks")
    if world == 1:
        return TilePlan(rank, world, width, height, 1, 1, 0, width, 0, height, 0, 0, 0, (0, width), ((0, height),))
    halo_rows = overlap + int(max_motion_rows) + 2 if gr > 1 else overlap
    halo_cols = overlap + int(max_motion_cols) + 2 if gc > 1 else overlap
    cell = cost_cell if cost is not None else 1
    mcol = None
    if cost is not None:
        import numpy as np
        cost = np.asarray(cost, np.uint64)
        ccols, crows = (width + cost_cell - 1) // cost_cell, (height + cost_cell - 1) // cost_cell
        if cost.shape[0] < crows or cost.shape[1] < ccols:
            raise ValueError("the cost map does not cover the image")
        cost = cost[:crows, :ccols]
        mcol = cost.sum(0).tolist()
    col_cuts = balanced_cuts(mcol, cell, width, gc, halo_cols if gc > 1 else 0)
    if col_cuts is None:
        raise ValueError(f"tiles of {width // gc} columns are thinner than the {halo_cols}-column history halo: use fewer GPUs or another grid")
    row_cuts = []
    for c in range(gc):            # every column of tiles cuts its own rows: by the cost inside it (a cell belongs to the column its first pixel column lies in)
        mrow = None
        if cost is not None:
            xs = np.arange(cost.shape[1]) * cost_cell
            mrow = cost[:, (xs >= col_cuts[c]) & (xs < col_cuts[c + 1])].sum(1).tolist()
        cuts = balanced_cuts(mrow, cell, height, gr, halo_rows if gr > 1 else 0)
        if cuts is None:
            raise ValueError(f"tiles of {height // gr} rows are thinner than the {halo_rows}-row history halo: use fewer GPUs or another grid")
        row_cuts.append(cuts)
    row_cuts = tuple(row_cuts)
    tr, tc = rank // gc, rank % gc
    return TilePlan(rank, world, width, height, gr, gc, col_cuts[tc], col_cuts[tc + 1], row_cuts[tc][tr], row_cuts[tc][tr + 1], overlap, halo_rows, halo_cols, col_cuts, row_cuts)


def refine_cost_map(cost, plans, times, cost_cell=8):
    """The feedback step of the cost-balanced planner: inside every rank's rectangle the map is scaled by (the rank's share of the ranks' summed frame
    times) / (its share of the map), so the next make_tile_plan(cost=...) moves the cuts towards the ranks that took longer than the map said.  What a
    running system has for free -- `world` floats to all-gather -- corrects what the ray kernels' wave lifetimes do not see (the SVGF pass, launches too
    small to fill the chip).  Two or three rounds bring busiest / mean from 1.15 to 1.01 on BASELINE config 5 at N = 8 (profiles/r6_tile_balance.txt).
    Returns a uint32 map on the same cells (rescaled to use the 32 bits; only ratios matter)."""
    import numpy as np
    c = np.asarray(cost, np.float64)
    total_t, total_c = float(sum(times)), float(c.sum())
    out = c.copy()
    for p, t in zip(plans, times):
        y0, y1, x0, x1 = p.row_begin // cost_cell, -(-p.row_end // cost_cell), p.col_begin // cost_cell, -(-p.col_end // cost_cell)
        share = float(c[y0:y1, x0:x1].sum()) / max(1e-30, total_c)
        out[y0:y1, x0:x1] = c[y0:y1, x0:x1] * ((t / max(1e-30, total_t)) / max(1e-9, share))
    return np.minimum(out * (6.0e7 / max(1e-30, out.max())), 6.0e7).astype(np.uint32)


def replan_transfers(old, new):
    """[(peer, send rect or None, recv rect or None)] that carry the path's cross-frame state -- temporal history, moments history, previous normals
    (hybrid_render_path.cpp:247-262) -- from the rectangles of `old`'s grid to those of `new`'s (a re-plan between two frames: the grid cut again
    at equal cost from the ranks' measured times, refine_cost_map).  Under the old plan a rank's values are exact where it OWNED the pixel, so
    recv = the pixels of my new rectangle grown by the new halo that `peer` owned, send = the mirror image; what I owned myself stays where it is.
    The old rectangles tile the image, so together with my own the receives cover everything the next frame's svgf.comp reads."""
    assert (old.world, old.rank, old.width, old.height) == (new.world, new.rank, new.width, new.height)
    dx, dy = (new.halo_cols if new.grid_cols > 1 else 0), (new.halo_rows if new.grid_rows > 1 else 0)
    need = [_grown(new.tile_rect(r), dx, dy, new.width, new.height) for r in range(new.world)]
    out = []
    for peer in range(new.world):
        if peer == new.rank:
            continue
        recv, send = _intersect(need[new.rank], old.tile_rect(peer)), _intersect(need[peer], old.tile_rect(old.rank))
        if recv or send:
            out.append((peer, send, recv))
    return out


def move_state(dist, tensors, old, new, group=None):
    """The re-plan's transfer (replan_transfers) of each [H, W, ...] tensor, in place, blocking (in stream order for device tensors): one grouped batch
    of point-to-point operations like every other exchange here.  Afterwards the tensors hold exact values on `new`'s rectangle grown by its halo."""
    pending = start_exchange(dist, tensors, new, None, group, rects=replan_transfers(old, new))
    if pending is not None:
        pending.finish()


def _rects_of(plan, margin, width):
    """The rectangle exchanges of a plan for a margin: `margin` is n_rows for a StripPlan, (halo_rows, halo_cols) or n for a TilePlan."""
    if isinstance(plan, TilePlan):
        hr, hc = margin if isinstance(margin, tuple) else (margin, margin)
        return plan.rect_exchanges(hr, hc)
    return [(peer, (0, width, sa, sb), (0, width, ra, rb)) for peer, (sa, sb), (ra, rb) in plan.exchanges(margin)]


def _view(t, rect):
    x0, x1, y0, y1 = rect
    return t[y0:y1] if (x0 == 0 and x1 == t.shape[1]) else t[y0:y1, x0:x1]


class PendingExchange:
    """A started neighbour exchange; finish() makes the current stream (or the host, for staged gloo) wait for it."""

    def __init__(self, reqs, staged, keep):
        self.reqs, self.staged, self.keep = reqs, staged, keep

    def finish(self):
        for req in self.reqs:
            req.wait()
        for dst, host in self.staged:
            dst.copy_(host)
        self.reqs, self.staged, self.keep = [], [], []


def start_exchange(dist, tensors, plan, n_rows, group=None, rects=None):
    """Neighbour halo exchange of a margin of each [H, W, ...] tensor (device or host), in place; `n_rows` is the number of rows for a
    StripPlan, (halo_rows, halo_cols) or one number for a TilePlan.  Returns a PendingExchange (or None if there is nothing to exchange).

    One grouped batch of point-to-point ops (ncclGroupStart/End under the "nccl" = RCCL backend): each
    neighbour pair talks over its direct xGMI link; no collective involves more than two ranks."""
    if plan.world == 1 or (rects is None and not n_rows):
        return None
    # Device tensors travel device-to-device under "nccl" (RCCL).  Under "gloo" (CPU transport: the CI route
    # for exercising this code with several ranks on one GPU) they are staged through host memory.
    stage = dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors)
    ops = []
    keep = []
    staged = []
    for peer, send_rect, recv_rect in (rects if rects is not None else _rects_of(plan, n_rows, tensors[0].shape[1])):      # (rects: a list of its own -- move_state)
        for t in tensors:
            if send_rect:
                send = _view(t, send_rect)
                send = send.cpu() if stage else (send if send.is_contiguous() else send.contiguous())     # a column range: packed
                ops.append(dist.P2POp(dist.isend, send, peer, group=group))
                keep.append(send)      # keep the send buffers alive until completion
            if recv_rect:
                recv = _view(t, recv_rect)
                if stage:
                    host = recv.cpu()
                    staged.append((recv, host))
                    recv = host
                elif not recv.is_contiguous():
                    tmp = recv.new_empty(recv.shape)
                    staged.append((recv, tmp))      # unpacked when the exchange is finished
                    recv = tmp
                ops.append(dist.P2POp(dist.irecv, recv, peer, group=group))
    if not ops:
        return None
    return PendingExchange(dist.batch_isend_irecv(ops), staged, keep)


class PreparedExchange:
    """The P2P descriptors of one recurring device-to-device exchange (same tensors, same rectangles every frame), built once:
    start() is then a single batch_isend_irecv call -- per frame the host only pays for the grouped launch, not for slicing
    tensors and constructing the P2POps (a 1080p frame on 8 GPUs leaves the host ~0.2 ms per frame in total).  Column ranges go
    through persistent contiguous staging tensors: packed at start(), unpacked when the exchange is finished."""

    def __init__(self, dist, tensors, plan, n_rows, group=None):
        self.dist = dist
        self.ops, self.keep, self.pack, self.unpack = [], [], [], []
        if plan.world == 1 or not n_rows:
            return
        if dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors):
            raise ValueError("prepared exchanges are device-to-device (RCCL); the staged gloo route goes through start_exchange")
        for peer, send_rect, recv_rect in _rects_of(plan, n_rows, tensors[0].shape[1]):
            for t in tensors:
                if send_rect:
                    send = _view(t, send_rect)
                    if not send.is_contiguous():
                        tmp = send.new_empty(send.shape)
                        self.pack.append((tmp, send))
                        send = tmp
                    self.ops.append(dist.P2POp(dist.isend, send, peer, group=group))
                    self.keep.append(send)
                if recv_rect:
                    recv = _view(t, recv_rect)
                    if not recv.is_contiguous():
                        tmp = recv.new_empty(recv.shape)
                        self.unpack.append((recv, tmp))
                        recv = tmp
                    self.ops.append(dist.P2POp(dist.irecv, recv, peer, group=group))
                    self.keep.append(recv)

    def start(self):
        if not self.ops:
            return None
        for tmp, src in self.pack:
            tmp.copy_(src, non_blocking=True)
        return PendingExchange(self.dist.batch_isend_irecv(self.ops), list(self.unpack), self.keep)


class StripGather:
    """C2 of SURVEY.md section 8e: the owners' rectangles of a per-rank image (the denoised shadow/AO image) assembled into one
    full frame on `root` -- what the display GPU's composition stage consumes.  Point-to-point like the halo exchanges (each
    rank sends its rectangle straight to the root over its direct xGMI link; tiles need not be equal), descriptors built
    once and replayed every frame, started after the SVGF pass and finished before the next frame's SVGF pass rewrites the
    image, i.e. it overlaps the next frame's ray tracing.  Under gloo with device tensors (CI route) the data is staged
    through host memory; column ranges (tiles) go through contiguous staging tensors either way."""

    def __init__(self, dist, image, plan, root=0, group=None):
        import torch
        self.dist, self.plan, self.root, self.image = dist, plan, root, image
        self.staged = dist.get_backend(group) == "gloo" and image.is_cuda
        self.group = group
        self.full = None
        if plan.world == 1:
            return
        H, W = image.shape[0], image.shape[1]
        if isinstance(plan, TilePlan):
            self.rects = [plan.tile_rect(r) for r in range(plan.world)]
        else:
            self.rects = [(0, W) + strip_bounds(plan.height, plan.world, r) for r in range(plan.world)]
        self.mine = self.rects[plan.rank]
        if plan.rank == root:
            self.full = torch.empty_like(image, device="cpu" if self.staged else image.device)

    def start(self):
        """Begin gathering the current contents of the image's owned rectangle; returns a PendingExchange (None for one rank)."""
        p = self.plan
        if p.world == 1:
            return None
        ops, keep, unpack = [], [], []
        if p.rank == self.root:
            for r, rect in enumerate(self.rects):
                if r == self.root:
                    continue
                dst = _view(self.full, rect)
                if not dst.is_contiguous():
                    tmp = dst.new_empty(dst.shape)
                    unpack.append((dst, tmp))
                    dst = tmp
                ops.append(self.dist.P2POp(self.dist.irecv, dst, r, group=self.group))
                keep.append(dst)
            _view(self.full, self.mine).copy_(_view(self.image, self.mine), non_blocking=True)      # the root's own rectangle: a local copy, in stream order
        else:
            send = _view(self.image, self.mine)
            send = send.cpu() if self.staged else (send if send.is_contiguous() else send.contiguous())
            ops.append(self.dist.P2POp(self.dist.isend, send, self.root, group=self.group))
            keep.append(send)
        return PendingExchange(self.dist.batch_isend_irecv(ops), unpack, keep)


def exchange_rows(dist, tensors, plan, n_rows, group=None):
    """Blocking form of start_exchange (the exchange is complete, in stream order, when this returns)."""
    pending = start_exchange(dist, tensors, plan, n_rows, group)
    if pending is not None:
        pending.finish()


class StripExchanges:
    """One rank's per-frame communication, as hooked into the two pass epilogues of the render graph (harness.HybridFrameLoop) --
    kept here, transport-agnostic, so that tests/test_tiling_gloo.py drives the very same ordering on CPU tensors over gloo:

        Raytrace Pass epilogue   after_raytrace():  the PREVIOUS frame's exchange #2 and gather have been in flight behind this
                                 frame's ray tracing; they land now (stream wait), before svgf.comp reads the history halo.
                                 With trace_overlap off, exchange #1 (E rows of raw visibility) happens here, blocking.
        SVGF pass epilogue       after_svgf():  start the gather (C2) of this frame's denoised rows and exchange #2 (history +
                                 moments halo rows for the NEXT frame's svgf.comp); neither is waited for.

    Descriptors are built once per distinct buffer set and replayed (PreparedExchange / StripGather, keyed by data pointers: the
    moments history alternates between two buffers, and with frames in flight the denoised image has one instance per slot).
    Device tensors under gloo (several ranks sharing one GPU in CI) are staged through the host every frame instead.
    A transport failure is an error unless `allow_degraded`, in which case the fall-back taken is listed in `degraded`."""

    def __init__(self, dist, plan, trace_overlap=True, denoise=True, gather=True, allow_degraded=False, group=None):
        self.dist, self.plan, self.group = dist, plan, group
        self.trace_overlap, self.denoise = bool(trace_overlap), bool(denoise)
        self.gather = bool(gather) and plan.world > 1 and self.denoise
        self.allow_degraded = bool(allow_degraded)
        self.degraded, self.gather_error = [], None
        self._prepared, self._gathers = {}, {}
        self._use_prepared = True
        self._pending = self._pending_gather = None
        self._last_gather = None

    def _halo(self):
        return (self.plan.halo_rows, self.plan.halo_cols) if isinstance(self.plan, TilePlan) else self.plan.halo

    def finish_pending(self):
        if self._pending is not None:
            self._pending.finish()
            self._pending = None
        if self._pending_gather is not None:
            self._pending_gather.finish()
            self._pending_gather = None

    def gathered_frame(self):
        """Root rank: the full frame assembled by the last finished gather, else None."""
        return self._last_gather.full if self._last_gather is not None else None

    def after_raytrace(self, raytraced=None):
        self.finish_pending()
        if self.trace_overlap or not self.denoise or self.plan.world == 1:
            return
        exchange_rows(self.dist, [raytraced() if callable(raytraced) else raytraced], self.plan, self.plan.overlap, self.group)   # exchange #1 (overlap on every cut side)

    def _fallback(self, what, e):
        if not self.allow_degraded:
            raise e
        self.degraded.append(f"{what}: {e!r}")
        import sys
        print(f"[strips] {what}: {e!r}", file=sys.stderr, flush=True)

    def after_svgf(self, denoised, history, moments):
        if self.plan.world == 1 or not self.denoise:
            return
        staged = self.dist.get_backend(self.group) == "gloo" and history.is_cuda
        if self.gather:                                   # C2
            try:
                g = self._gathers.get(denoised.data_ptr())
                if g is None:
                    g = self._gathers[denoised.data_ptr()] = StripGather(self.dist, denoised, self.plan, group=self.group)
                self._last_gather = g
                self._pending_gather = g.start()
            except Exception as e:   # noqa: BLE001
                self.gather, self._last_gather, self.gather_error = False, None, repr(e)
                self._fallback("strip gather disabled", e)
        if self._use_prepared and not staged:             # exchange #2, replayed descriptors
            try:
                key = (history.data_ptr(), moments.data_ptr())
                prepared = self._prepared.get(key)
                if prepared is None:
                    prepared = self._prepared[key] = PreparedExchange(self.dist, [history, moments], self.plan, self._halo(), self.group)
                self._pending = prepared.start()          # consumed by the NEXT frame
                return
            except Exception as e:   # noqa: BLE001
                self._use_prepared = False
                self._fallback("prepared exchange disabled (descriptors rebuilt every frame)", e)
        self._pending = start_exchange(self.dist, [history, moments], self.plan, self._halo(), self.group)
