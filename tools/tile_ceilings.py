"""Compute-only ceilings of an N-GPU run, measured on ONE GPU with virtual tiles (no exchanges), and the per-tile-size tuning sweep behind
harness.TILE_TUNING.  For N in {1, 2, 4, 8} the rank whose rectangle computes the most pixels is set with vhr_set_tile and the
frame (Raytrace Pass + SVGF Denoise Pass) is timed by wall clock: what an N-GPU run cannot beat before any byte is exchanged.

   python tools/tile_ceilings.py <config2|config3|config4|config5> [--fif 1|2] [--sweep] [--tuned] [option=value ...]

--sweep: at every N the options that can matter for a thin launch are toggled ONE AT A TIME against the defaults, the ones that gain more than
         1 % are combined and re-measured (one JSON line per arm; the winners go to harness.TILE_TUNING by tile area)
--tuned: apply harness.tuned_tile_options(computed pixels) at every N (what HybridFrameLoop does for world > 1)
--balance: (round 6) instead of the ceilings: at N = 4 and 8, EVERY rank's frame time on the equal-pixel grid and on the grid cut at equal cost
         (vhr_tile_plan_make_weighted on the whole-image frame's wave lifetimes, Context.tile_cost_map()): busiest, mean, busiest / mean -- what a
         cost-balanced planner takes off the slowest rank
Also prints, per N, the projection of DESIGN.md section 5: exchange bytes of the busiest rank (history + moments halo, gather share), their
time at one xGMI link's 153 GB/s, and ceiling + that time (exposed) next to max(ceiling, that time) (overlapped behind the next frame's rays)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import harness, scenes, tiling
from vulkanhybridrenderer_amd.harness import HybridFrameLoop

CONFIGS = {"config2": ("sponza_proc", 1920, 1080, 2, 0), "config3": ("sponza_proc", 3840, 2160, 4, 0), "config4": ("bistro_proc", 1920, 1080, 2, 1),
           "config5": ("bistro_proc", 3840, 2160, 16, 2)}
SWEEP = [("fuse_temporal", 1), ("atrous_small_tiles", 0), ("atrous_small_tiles", 1), ("raygen_waves_per_block", 1), ("raygen_waves_per_block", 4),
         ("raygen_cost_order", 0), ("raygen_cost_order", 2), ("svgf_async_unread", 0), ("svgf_async_unread", 2), ("raygen_tile_rows", 6), ("raygen_tile_rows", 8),
         ("reflection_async", 0)]
if os.environ.get("VHR_SWEEP"):                     # another list of arms: VHR_SWEEP="raygen_tile_rows=4,refill_threshold=8"
    SWEEP = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in os.environ["VHR_SWEEP"].split(",")]
ONLY_N = [int(v) for v in os.environ["VHR_ONLY_N"].split(",")] if os.environ.get("VHR_ONLY_N") else None
XGMI_GBS, GROUP_LATENCY_US = 153.0, 20.0          # one direct link between two GPUs of the node; one grouped RCCL batch (MI355X_MICROARCH.md)
args = [a for a in sys.argv[1:] if "=" not in a and not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
name = args[0]
FIF = int(sys.argv[sys.argv.index("--fif") + 1]) if "--fif" in sys.argv else 1
if "--fif" in sys.argv:
    args = [a for a in args if a != str(FIF)]
scene_name, W, H, ao, refl = CONFIGS[name]
scene = getattr(scenes, scene_name)()
NF = 12
loop = HybridFrameLoop(scene, W, H, NF, shadow=True, ao_spp=ao, reflections=refl, denoise=True, frames_in_flight=FIF)
defaults = {}
for k, v in opts:
    loop.ctx.set_option(k, int(v))


def measure(plan, n, extra=(), reps=3, frames=8):
    loop.ctx.set_tile(plan.col_begin, plan.col_end, plan.row_begin, plan.row_end, plan.overlap, plan.halo_rows, plan.halo_cols)
    loop.ctx.set_option("trace_overlap", 1 if n > 1 else 0)
    loop.ctx.set_option("strip_shrink_overlap", 1 if n > 1 else 0)
    loop.ctx.set_option("reflection_async", 2 if n > 1 else 1)           # what the multi-GPU harness sets
    saved = {k: loop.ctx.get_option(k) for k, _ in extra}
    for k, v in extra:
        loop.ctx.set_option(k, v)
    for i in range(3):
        loop.frame(i)
    torch.cuda.synchronize(); loop.ctx.synchronize()
    ts = []
    for rep in range(reps):
        t0 = time.perf_counter()
        for i in range(3, 3 + frames):
            loop.frame(i)
        torch.cuda.synchronize(); loop.ctx.synchronize()
        ts.append((time.perf_counter() - t0) / frames * 1e3)
    for k, v in saved.items():
        loop.ctx.set_option(k, v)
    return float(min(ts))


def exchange_bytes(plan):
    """Bytes the rank receives per frame: the history (8 B/px) + moments (4 B/px) halo from every peer, and -- rank 0 -- the other tiles of Denoised."""
    halo = sum((r[1] - r[0]) * (r[3] - r[2]) for _, _, r in plan.rect_exchanges(plan.halo_rows, plan.halo_cols) if r) * 12
    gather = (W * H - (plan.col_end - plan.col_begin) * (plan.row_end - plan.row_begin)) * 8 if plan.rank == 0 else 0
    peers = len([1 for _, _, r in plan.rect_exchanges(plan.halo_rows, plan.halo_cols) if r])
    return halo, gather, peers


area = lambda p: (p.computed_rect()[1] - p.computed_rect()[0]) * (p.computed_rect()[3] - p.computed_rect()[2])
if "--balance" in flags:
    loop.ctx.set_option("raygen_cost_order", 2)            # every launch leaves its wave lifetimes
    for i in range(6):
        loop.frame(i)
    torch.cuda.synchronize(); loop.ctx.synchronize()
    rays = loop.ctx.tile_cost_map()
    # what the map does not see: the SVGF pass, whose cost goes by pixels -- added as a constant per cell, in the pass times' proportion
    loop.ctx.gather_performance_statistics()
    rt_ms, svgf_ms = loop.ctx.pass_time_ms("Raytrace Pass")[1], loop.ctx.pass_time_ms("SVGF Denoise Pass")[1]
    per_cell = float(rays.sum(dtype=np.float64)) / rays.size * (svgf_ms / max(1e-9, rt_ms))
    cost = (rays.astype(np.float64) + per_cell).astype(np.uint32)
    whole = tiling.make_tile_plan(W, H, 1, 0)
    t1 = measure(whole, 1)
    print(json.dumps({"config": name, "n": 1, "ms_per_frame": round(t1, 4), "pass_ms": [round(rt_ms, 4), round(svgf_ms, 4)], "cost_map": {"cells": list(cost.shape), "rays_max_over_mean": round(float(rays.max() / max(1.0, rays.mean())), 2), "per_cell_constant_for_the_svgf_pass": round(per_cell)}}), flush=True)
    for n in (4, 8):
        refined = cost
        arms = [("equal pixels", None), ("equal cost", cost)] + [(f"equal cost, refined by the ranks' measured times ({k})", "refine") for k in (1, 2, 3)]
        for label, c in arms:
            if isinstance(c, str):
                # what a running system has for free: every rank's frame time (8 floats to all-gather).  Inside each rank's rectangle the map is scaled by
                # (the rank's share of the summed times) / (its share of the map): the next plan moves the cuts towards the ranks that took longer than their cost said
                refined = tiling.refine_cost_map(refined, plans, ts)
                c = refined
            try:
                plans = [tiling.make_tile_plan(W, H, n, r, loop.max_motion_rows, loop.max_motion_cols, grid=None, cost=c) for r in range(n)]
            except ValueError as e:
                print(f"{name} N={n} {label}: {e}", flush=True); continue
            ts = [measure(p, n) for p in plans]
            full = np.kron(cost.astype(np.float64) / 64.0, np.ones((8, 8)))[:H, :W]
            shares = [float(full[p.row_begin:p.row_end, p.col_begin:p.col_end].sum() / full.sum()) for p in plans]
            print(json.dumps({"config": name, "n": n, "plan": label, "grid": f"{plans[0].grid_rows}x{plans[0].grid_cols}", "col_cuts": list(plans[0].col_cuts), "row_cuts": [list(rc) for rc in plans[0].row_cuts],
                              "ms_per_rank": [round(t, 4) for t in ts], "busiest_ms": round(max(ts), 4), "mean_ms": round(float(np.mean(ts)), 4), "busiest_over_mean": round(max(ts) / float(np.mean(ts)), 3),
                              "share_of_linear_pct_by_the_busiest": round(100.0 * t1 / (n * max(ts)), 1), "cost_share_of_the_busiest_rank": round(max(shares), 4), "cost_share_ideal": round(1.0 / n, 4)}), flush=True)
    loop.close()
    sys.exit(0)
base = None
for n in (1, 2, 4, 8):
    try:
        plans = [tiling.make_tile_plan(W, H, n, r, loop.max_motion_rows, loop.max_motion_cols, grid=None) for r in range(n)]
    except ValueError as e:
        print(f"{name} N={n}: {e}", flush=True); continue
    plan = max(plans, key=area)                       # the rank that computes the most pixels
    tuned = tuple(harness.tuned_tile_options(area(plan), n).items()) if "--tuned" in flags and n > 1 else ()
    ms = measure(plan, n, tuned)
    base = ms if n == 1 else base
    c = plan.computed_rect()
    line = {"config": name, "frames_in_flight": FIF, "n": n, "grid": f"{plan.grid_rows}x{plan.grid_cols}", "owned": [plan.col_end - plan.col_begin, plan.row_end - plan.row_begin],
            "computed": [c[1] - c[0], c[3] - c[2]], "computed_pixels": area(plan), "extra_pixels_pct": round(100.0 * area(plan) * n / (W * H) - 100.0, 1),
            "options": dict(tuned) or None, "ms_per_frame": round(ms, 4), "share_of_linear_pct": round(100.0 * base / (n * ms), 1)}
    if n > 1:
        # the projection: rank 0 receives the gather (every other tile over its own link, in parallel) and its halo; the exchange is one grouped batch
        halo, gather, peers = exchange_bytes(plans[0])
        per_link = max(gather / max(1, n - 1), halo / max(1, peers)) if n > 1 else 0
        comm_ms = (per_link / (XGMI_GBS * 1e9) * 1e3) + GROUP_LATENCY_US * 1e-3
        line["projection"] = {"halo_bytes": halo, "gather_bytes_rank0": gather, "bytes_on_the_busiest_link": int(per_link), "exchange_ms": round(comm_ms, 4),
                              "ms_exposed": round(ms + comm_ms, 4), "ms_overlapped": round(max(ms, comm_ms), 4),
                              "share_of_linear_exposed_pct": round(100.0 * base / (n * (ms + comm_ms)), 1), "share_of_linear_overlapped_pct": round(100.0 * base / (n * max(ms, comm_ms)), 1)}
    print(json.dumps(line), flush=True)
    if "--sweep" in flags and n > 1 and (ONLY_N is None or n in ONLY_N):
        gains = []
        for k, v in SWEEP:
            if (k == "reflection_async" and not refl) or loop.ctx.get_option(k) == v:
                continue
            t = measure(plan, n, ((k, v),))
            print(json.dumps({"config": name, "n": n, "arm": f"{k}={v}", "ms_per_frame": round(t, 4), "vs_default_pct": round(100.0 * (t / ms - 1.0), 2)}), flush=True)
            if t < 0.99 * ms:
                gains.append((t, k, v))
        best = {}
        for t, k, v in sorted(gains):                  # the best value of each option that gained
            best.setdefault(k, v)
        if best:
            t = measure(plan, n, tuple(best.items()))
            print(json.dumps({"config": name, "n": n, "arm": "combined " + ",".join(f"{k}={v}" for k, v in best.items()), "ms_per_frame": round(t, 4),
                              "vs_default_pct": round(100.0 * (t / ms - 1.0), 2), "share_of_linear_pct": round(100.0 * base / (n * t), 1)}), flush=True)
loop.close()
