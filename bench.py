#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hybrid ray-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s + ms/frame, "Sponza 1080p RT shadows + AO + SVGF" at 1/2/4/8 GPUs.
A step = one frame of the hot path (Raytrace Pass + SVGF Denoise Pass, reference schedule: 1 temporal + 5
a-trous dispatches + 3 blits) over a G-buffer already resident in HBM.  Sponza itself is not available
(SURVEY.md section 8d): the workload is the procedural `sponza_proc` atrium (257 536 triangles, 103 primitives)
with the camera dolly of section 8(d); data = synthetic.

    python bench.py --gpus 1 --steps 32 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = unique rays traced by all ranks / wall time of the K timed frames
(max over ranks).  Extra objects: `roofline` (the a-trous kernel against the 8 TB/s HBM peak, timed live with HIP
events on the launch stream), `cpu_baseline` (the CPU oracle -- a restatement, NOT lavapipe -- on the host cores
over a bounded sample), `traversal` (Mrays/s of the ray-tracing kernel alone), `passes` (per-pass ms under the
reference's pass names).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate -- the level the (cache-resident) tree is read from
ATROUS_BYTES_PER_PIXEL = 24      # read integrated 8 + normals/id 8, write 8
TEMPORAL_BYTES_PER_PIXEL = 52    # normals 8 + motion 8 + current 4 + previous normals 8 + history 8 + moments 4, integrated 8 + moments 4 out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=32)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--ao-spp", type=int, default=2)
    p.add_argument("--reflections", action="store_true", help="also trace the mirror ray raygen.rgen always issues")
    p.add_argument("--refl-bounces", type=int, default=None, choices=[0, 1, 2],
                   help="mirror bounces (1 = --reflections = the reference; 2 = BASELINE config 5's extension; second-bounce rays are NOT counted in value)")
    p.add_argument("--scene", default="sponza_proc", choices=["sponza_proc", "bistro_proc", "sponza_hard", "tiny", "sponza_proc_rot", "sponza_hard_rot", "bistro_proc_rot"],
                   help="*_rot: the same scene with the whole world (geometry, light, camera path) turned off the world axes (scenes.rotated)")
    p.add_argument("--bvh-frame", type=int, default=None, choices=[0, 1], help="option \"bvh_frame\": 1 (the library's default) = the boxes in the frame the builder finds for the scene (csrc/bvh_frame.hpp), 0 = along the world axes")
    p.add_argument("--bvh-presplit", type=int, default=0, help="option \"bvh_presplit\": budget of extra triangle references in percent (csrc/presplit.hpp), 0 = off")
    p.add_argument("--gltf", default=None, help="load this .gltf / .glb instead of a procedural scene (vulkanhybridrenderer_amd/gltf.py)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-frames", type=int, default=16, help="frames of the same workload the CPU oracle is timed on (~1.3 s each on the GPU box's 128 threads)")
    p.add_argument("--max-gbuffers", type=int, default=64, help="distinct precomputed G-buffer frames (wraps beyond)")
    p.add_argument("--replan-frame", type=int, default=None,
                   help="N > 1: after this verified frame the grid is cut again at equal cost -- a uniform map scaled by the ranks' measured "
                        "frame times (tiling.refine_cost_map; VHR_BENCH_REPLAN_TIMES=t0,t1,... overrides them) -- and the SVGF history follows its pixels "
                        "(HybridFrameLoop.replan); the remaining verified frames check the NEW rectangles against the single context")
    p.add_argument("--verify-frames", type=int, default=3,
                   help="N > 1 only: before timing, check that the gathered strips equal a single full-frame context bit for bit")
    p.add_argument("--backend", default=os.environ.get("VHR_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                   help="gloo + --share-device lets several ranks run on ONE GPU (functional check of the strip path only)")
    p.add_argument("--exchange-raytraced", action="store_true",
                   help="N > 1: trace owned rows only and fetch the overlap rows' raw shadow/AO from the neighbours "
                        "(default: every rank also traces its 30 overlap rows; no exchange on the critical path)")
    p.add_argument("--no-gather", action="store_true",
                   help="N > 1: leave the denoised strips on their GPUs (default: gathered to rank 0 every frame, asynchronously, "
                        "inside the timed region -- the frame the composition stage of the display GPU consumes)")
    p.add_argument("--share-device", action="store_true", default=bool(os.environ.get("VHR_BENCH_SHARE_DEVICE")))
    p.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3],
                   help="frames in flight of the timed region (vhr_set_option frames_in_flight; 1 = the single-stream contract). "
                        "At N = 1 the line carries the 2-frames-in-flight figures as extra fields either way")
    p.add_argument("--min-seconds", type=float, default=1.0,
                   help="the timed block of --steps frames is repeated until this much time has been measured; value = the median block")
    p.add_argument("--no-extras", action="store_true", help="skip the extra blocks (mirror-ray frame, frames in flight) of the N = 1 line")
    p.add_argument("--grid", default="auto", help="N > 1: the screen decomposition -- auto = the planner's grid of screen tiles (2x4 at N = 8: the busiest "
                   "rank computes +19 %% instead of a row strip's +44 %%), strips = row strips, or ROWSxCOLS")
    p.add_argument("--comm", default="auto", choices=["auto", "torch", "c_abi"],
                   help="N > 1: who moves the halos and the gather -- torch = torch.distributed P2P (tiling.py), c_abi = the library's own RCCL calls "
                        "(vhr_comm_*, csrc/comm.cpp), auto = c_abi on the nccl backend if it comes up and reproduces the single-context frame, else torch")
    p.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                   help="vhr_set_option KEY VALUE on the context before the first frame (A-B runs and profiles; every option is result-neutral). "
                        "Anything set this way is listed under config.options")
    p.add_argument("--allow-degraded", action="store_true",
                   help="N > 1: fall back (per-frame descriptors / no gather) instead of failing when the transport refuses the replayed exchanges")
    p.add_argument("--print-workload-key", action="store_true", help="print the name profiles/pmc_<key>.json goes by for this workload and exit (tools/pmc_workload.sh)")
    p.add_argument("--dry-launch", action="store_true",
                   help="--gpus N > 1 without WORLD_SIZE: print the torch.distributed.run command the launcher would start (one JSON line) and exit")
    p.add_argument("--no-c-abi-probe", action="store_true",
                   help="launcher only: skip the second, time-bounded child that repeats a short run through the library's own RCCL calls (--comm c_abi)")
    p.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher only: seconds the ranks may take before the launcher ends them")
    args = p.parse_args()
    if args.share_device and args.backend == "nccl":
        args.backend = "gloo"            # RCCL refuses two ranks on one device: ranks that share a GPU talk over gloo
    return args


def visible_devices():
    """GPUs this process may use.  torch.cuda.device_count() reads the driver's device list without creating a HIP context
    (the launcher must never touch the GPU: its children do)."""
    import torch
    return int(torch.cuda.device_count())


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _error_line(message, args, **extra):
    print(json.dumps({"error": message, "n_gpus": args.gpus, "metric": "Mrays/s (unique rays) + ms/frame, Sponza 1080p RT shadows+AO+SVGF",
                      "value": None, **extra}), flush=True)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: this process becomes the launcher.  It starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process, relays rank 0's
    JSON line and returns the child's exit code.  It makes no HIP call itself.  With fewer than N devices visible (and no
    --share-device) it prints an error line and exits 2: it never falls through to a one-GPU measurement."""
    import subprocess
    n = args.gpus
    have = visible_devices()

    def command(extra):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"] + extra
    cmd = command([])
    enough = have >= 1 and (have >= n or args.share_device)
    if args.dry_launch:
        print(json.dumps({"dry_launch": cmd, "n_gpus": n, "devices_visible": have, "would_run": bool(enough), "share_device": bool(args.share_device),
                          "backend": args.backend}), flush=True)
        return 0
    if not enough:
        _error_line((f"--gpus {n} but {have} device(s) visible; --share-device runs {n} ranks on one GPU over gloo (a functional check, not a measurement)")
                    if have >= 1 else "no GPU visible (the product has no CPU path)", args, devices_visible=have)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))

    def run(cmd, timeout):
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=timeout)
            return p.returncode, out, False
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(p.pid, signal.SIGTERM)          # exactly the process group started above
                out, _ = p.communicate(timeout=20)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                out, _ = p.communicate()
            return 124, out, True

    def json_line(out):
        for l in reversed((out or "").splitlines()):
            if l.startswith("{"):
                try:
                    return json.loads(l)
                except ValueError:
                    continue
        return None
    rc, out, timed_out = run(cmd, args.launch_timeout)
    line = json_line(out)
    if line is None:
        sys.stdout.write(out or "")
        _error_line("the ranks ended without a result line" + (" (launcher timeout)" if timed_out else ""), args, child_exit_code=rc)
        return rc or 1
    # The measured route is torch.distributed (backend nccl == RCCL).  The library's own RCCL calls (vhr_comm_*, csrc/comm.cpp) get a
    # second, short, time-bounded run of their own so that a fault in a route that has never met a second device cannot cost the
    # measurement above: its outcome is reported as `c_abi_route`, never as `value`.
    if (rc == 0 and "error" not in line and not args.no_c_abi_probe and args.comm == "auto" and args.backend == "nccl" and not args.share_device):
        probe = command(["--comm", "c_abi", "--no-extras", "--no-cpu-baseline", "--min-seconds", "0.3"])
        prc, pout, pto = run(probe, 240.0)
        pl = json_line(pout)
        if prc == 0 and pl and "error" not in pl:
            line["c_abi_route"] = {"exchanges_through": pl["config"]["exchanges_through"], "ms_per_step": pl["ms_per_step"], "value": pl["value"],
                                   "strips_vs_single_context": pl["config"]["strips_vs_single_context"], "exchanges_note": pl["config"].get("exchanges_note")}
        else:
            line["c_abi_route"] = {"error": (pl or {}).get("error", "timed out after 240 s" if pto else f"exit code {prc}"), "exit_code": prc}
    print(json.dumps(line), flush=True)
    return rc


def _bounces(args):
    return args.refl_bounces if args.refl_bounces is not None else int(args.reflections)


def cpu_baseline(scene, W, H, tp, n_frames, rays_per_pixel):
    """The oracle (CPU restatement, not lavapipe) over the first `n_frames` frames of the same workload."""
    from oracle import binding as ob
    from vulkanhybridrenderer_amd import camera
    ob.build()
    osc = ob.Scene(scene)
    svgf = ob.SVGF(W, H)
    pfds = camera.dolly_frames(scene, W, H, n_frames + 1)
    gbufs = [osc.gbuffer(p, W, H) for p in pfds[1:]]          # G-buffer production is outside the timed region
    rays = 0
    t0 = time.perf_counter()
    for pfd, g in zip(pfds[1:], gbufs):
        sa, _, _, r = osc.raygen(pfd, tp, g[0], g[2], want_reflections=bool(tp["reflections"]))
        svgf.frame(pfd, g[0], g[1], sa)
        rays += r
    dt = time.perf_counter() - t0
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=ob.max_threads(), kind="port",
                sample=f"{n_frames} frames of the same {W}x{H} workload (trace + SVGF), OpenMP over rows; CPU restatement, not lavapipe",
                ms_per_frame=dt / n_frames * 1e3)


def verify_strips(args, scene, loop, dist, rank, world, device):
    """Strips gathered from all ranks == one full-frame context, bit for bit (same kernels, same inputs)."""
    import torch
    from vulkanhybridrenderer_amd import lib
    from vulkanhybridrenderer_amd.harness import HybridFrameLoop, alias_tensor
    W, H, V = args.width, args.height, args.verify_frames
    ref = HybridFrameLoop(scene, W, H, V, shadow=True, ao_spp=args.ao_spp, reflections=_bounces(args), denoise=True, device=device) if rank == 0 else None   # single stream
    ok = True
    cpu = args.backend == "gloo"
    for i in range(V):
        t_frame = time.perf_counter()
        loop.frame(i)
        loop.finish_pending_exchange()
        torch.cuda.synchronize()
        t_frame = time.perf_counter() - t_frame
        x0, x1, y0, y1 = loop.owned_rect()
        def tile_of(ctx_, image):
            t = alias_tensor(ctx_.transient_info(image)).view(torch.int16)
            return t.cpu() if cpu else t
        mine = tile_of(loop.ctx, lib.DENOISED)[y0:y1, x0:x1].contiguous()
        mine_refl = tile_of(loop.ctx, lib.REFLECTIONS)[y0:y1, x0:x1].contiguous() if _bounces(args) else None      # (not denoised: the mirror ray's own launch)
        if mine_refl is not None:
            mine = torch.cat([mine, mine_refl], dim=-1).contiguous()           # one message per rank: denoised | reflections
        sizes = [None] * world
        dist.all_gather_object(sizes, (x0, x1, y0, y1))
        if rank == 0:
            ref.frame(i)
            torch.cuda.synchronize()
            full = tile_of(ref.ctx, lib.DENOISED)
            if mine_refl is not None:
                full = torch.cat([full, tile_of(ref.ctx, lib.REFLECTIONS)], dim=-1)
            ok &= bool(torch.equal(mine, full[y0:y1, x0:x1]))
            gathered = loop.gathered_frame()                              # C2: the frame assembled on rank 0 (tiling.StripGather / vhr_comm_*)
            if gathered is not None:
                ok &= bool(torch.equal(gathered.view(torch.int16).cpu(), full[..., :4].cpu()))
            for r in range(1, world):
                a0, a1, b0, b1 = sizes[r]
                buf = torch.empty((b1 - b0, a1 - a0, full.shape[-1]), dtype=torch.int16, device=mine.device)
                dist.recv(buf, src=r)
                ok &= bool(torch.equal(buf, full[b0:b1, a0:a1]))
        else:
            dist.send(mine, dst=0)
        if args.replan_frame is not None and i == args.replan_frame:
            # the re-plan while frames run: what a running system has for free -- every rank's last frame time, all-gathered -- scales a cost map inside
            # the ranks' rectangles; the grid is cut again and the cross-frame state follows its pixels
            import numpy as np
            from vulkanhybridrenderer_amd import tiling
            times = [None] * world
            dist.all_gather_object(times, float(t_frame))
            if os.environ.get("VHR_BENCH_REPLAN_TIMES"):
                times = [float(v) for v in os.environ["VHR_BENCH_REPLAN_TIMES"].split(",")][:world]
            base = loop.tile_cost if loop.tile_cost is not None else np.full(((H + 7) // 8, (W + 7) // 8), 1000, np.uint32)
            old = loop.plan
            plans = [tiling.make_tile_plan(W, H, world, r, loop.max_motion_rows, loop.max_motion_cols, loop.atrous_steps, grid=(old.grid_rows, old.grid_cols), cost=loop.tile_cost)
                     for r in range(world)]
            new = loop.replan(tiling.refine_cost_map(base, plans, times))
            loop.replan_info = {"after_frame": i, "times_ms": [round(t * 1e3, 3) for t in times], "rect_before": list(old.rect), "rect_after": list(new.rect)}
    if ref is not None:
        ref.close()
    flag = torch.tensor([1 if ok else 0], device="cpu" if cpu else "cuda")
    dist.broadcast(flag, src=0)
    return "bit-identical" if int(flag[0]) else "MISMATCH"


SIMDS, CLOCK_HZ = 1024, 2.4e9     # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2400 MHz max clock


def workload_key(args):
    """What a PMC file is collected for: scene, size, rays.  One GPU only (the counters are per launch of the whole frame)."""
    scene = os.path.splitext(os.path.basename(args.gltf))[0] if args.gltf else args.scene
    return f"{scene}_{args.width}x{args.height}_ao{args.ao_spp}_refl{_bounces(args)}"


def load_pmc(args, fingerprint):
    """profiles/pmc_<workload>.json (tools/pmc_workload.sh): the counters behind roofline.traffic, roofline.valu and traversal.address_unit.
    Quoted only when the file was collected on THIS library (vhr_source_fingerprint) and for THIS workload at N = 1; else (None, why)."""
    if args.gpus != 1:
        return None, "PMC files are per whole-frame launch: not quoted for N > 1"
    name = f"pmc_{workload_key(args)}.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, f"no profiles/{name} (tools/pmc_workload.sh)"
    if d.get("fingerprint") != fingerprint:
        return None, f"profiles/{name} was collected on library {d.get('fingerprint')}; the loaded library is {fingerprint}: stale, not quoted"
    return d, None


def address_unit_block(pmc, raygen_ms):
    """What the ray-tracing kernel's memory path does: ~20 busy cycles of the CU's address unit (TA) per wave-level load instruction whatever its
    width or the number of active lanes.  `floor_us` is the time the launch needs with the address units of all 256 CUs busy every cycle."""
    ta = (pmc or {}).get("raygen_ta")
    if not ta:
        return None
    floor_us = ta["ta_busy_cycles_sum"] / 256.0 / CLOCK_HZ * 1e6
    return {"load_instructions_per_launch": ta["wave_level_load_instructions"], "ta_busy_cycles_per_load": ta["ta_cycles_per_load_instruction"],
            "ta_busy_frac_under_profiler": ta["ta_busy_frac"], "floor_us": round(floor_us, 1), "frac": round(floor_us / max(raygen_ms * 1e3, 1e-9), 4),
            "source": "rocprofv3 --pmc TA_TA_BUSY_sum, TA_FLAT_READ_WAVEFRONTS_sum (tools/pmc_workload.sh); floor = TA busy cycles / 256 CUs / 2.4 GHz"}


def reflection_block(ctx, loop, frame_index, sync):
    """`traversal_reflection`: the mirror-ray launch's own counters and time (reflection_queue_kernel: raygen.rgen:59-65 + reflection_hit.rchit).
    Eight frames with an event pair on the launch and "reflection_async" 0 -- the launch alone on the chip, which is what avg_launch_ms,
    mrays_per_s, effective_traversal_gbs and l2_frac describe --, then one untimed frame with the in-kernel statistics on (the library runs a
    statistics frame in order too).  With "reflection_async" on, the same launch's duration beside the SVGF pass is reported next to it."""
    async_mode = ctx.get_option("reflection_async")
    ctx.set_option("reflection_async", 0)
    for i in range(frame_index, frame_index + 2):          # untimed: the first launch of a kernel on a stream can grow the stream's scratch (milliseconds, once)
        loop.frame(i)
    sync()
    ctx.set_kernel_timing(["reflection"])
    ctx.kernel_time("reflection", reset=True)
    for i in range(frame_index, frame_index + 8):
        loop.frame(i)
    sync()
    ms, n = ctx.kernel_time("reflection")
    ctx.set_kernel_timing(False)
    ctx.set_option("reflection_async", async_mode)
    ctx.set_ray_statistics(True)
    loop.frame(frame_index + 8)
    sync()
    st = ctx.reflection_statistics()
    ctx.set_ray_statistics(False)
    if not n or not st["rays"]:
        return None
    launch_ms = ms / n
    visits, tests, rays = st["node_visits"], st["triangle_tests"], st["rays"]
    gbs = (visits * 48 + tests * 48) / launch_ms / 1e6            # 48-byte nodes (three loads per visit), 48-byte triangle records
    return {"kernel": "reflection_queue_kernel (closest-hit walk of the mirror rays on the 48-byte nodes + reflection_hit.rchit per tile)",
            "avg_launch_ms": round(launch_ms, 4), "rays": int(rays), "second_bounce_rays": int(st["second_bounce_rays"]),
            "mrays_per_s": round(rays / launch_ms / 1e3, 1),
            "node_visits_per_ray": round(visits / rays, 2), "triangle_tests_per_ray": round(tests / rays, 2), "leaf_visits_per_ray": round(st["leaf_visits"] / rays, 2),
            "active_lane_utilisation": round(st["active_lane_utilisation"], 3),
            "refills_per_wave": round(st["refills"] / max(1, st["waves"]), 2),
            "walk_share_of_wave_lifetime": round(st["cycles_walk"] / max(1, st["cycles_total"]), 3),
            "effective_traversal_gbs": round(gbs, 1), "l2_frac": round(gbs / L2_PEAK_GBS, 4),
            "shares_the_chip_with": None, "timed_with": "reflection_async 0 (the launch alone on the chip; the frame itself runs it beside the SVGF pass when the option is on)",
            "note": "utilisation = (node visits + triangle tests) / (64 x wave-level trips of those loops); l2_frac = effective_traversal_gbs / the L2's 34.5 TB/s "
                    "(MI355X_MICROARCH.md: the tree is cache resident, HBM is not the level it is read from)"}


def time_blocks(loop, barrier, first_frame, steps, min_seconds, max_blocks=400, slowest=None):
    """Blocks of exactly `steps` frames, each bracketed by barrier + synchronize, until `min_seconds` have been measured.
    `slowest(dt)`: with more than one rank, the MAX of the block's time over the ranks -- the stop decision is taken on that one number,
    so every rank runs the same count of blocks (a rank that stopped on its own clock would leave the others in a barrier: r4, one
    run in five at --min-seconds 0.05).  Returns (this rank's block seconds, next frame index)."""
    times, f, total = [], first_frame, 0.0
    while True:
        barrier()
        t0 = time.perf_counter()
        for i in range(f, f + steps):
            loop.frame(i)
        barrier()
        dt = time.perf_counter() - t0
        times.append(dt)
        total += slowest(dt) if slowest is not None else dt
        f += steps
        if total >= min_seconds or len(times) >= max_blocks:
            return times, f


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL's peer-to-peer setup needs on this driver (already exported on the GPU boxes)
    if args.print_workload_key:
        print(workload_key(args))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # not under torchrun: become the launcher (before anything touches the GPU)
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    if os.environ.get("VHR_BENCH_WATCHDOG"):                      # debugging aid: every rank dumps its Python stacks and exits after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VHR_BENCH_WATCHDOG"]), exit=True)
    if args.dry_launch:
        print(json.dumps({"dry_launch": None, "n_gpus": args.gpus, "note": "nothing to launch: one rank, or already under torch.distributed.run"}), flush=True)
        return
    import torch
    import torch.distributed as dist
    from vulkanhybridrenderer_amd import abi, lib, scenes
    from vulkanhybridrenderer_amd.harness import HybridFrameLoop

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            _error_line(f"--gpus {args.gpus} but WORLD_SIZE={world}", args)
        raise SystemExit(2)
    have = visible_devices()
    if have < 1 or (have < world and not args.share_device):      # never fall through to fewer GPUs than the line would claim
        if rank == 0:
            _error_line(f"{world} rank(s) but {have} device(s) visible (--share-device runs the ranks on one GPU over gloo: a functional check)", args, devices_visible=have)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product has no CPU path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.gltf:                                  # a real asset (Sponza.gltf, Bistro.glb ...) through the scene_loader.cpp-equivalent host
        from vulkanhybridrenderer_amd import gltf
        scene = gltf.load(args.gltf)
    else:
        scene = {"sponza_proc": scenes.sponza_proc, "bistro_proc": scenes.bistro_proc, "sponza_hard": scenes.sponza_hard, "tiny": scenes.tiny_scene,
                 "sponza_proc_rot": scenes.sponza_proc_rot, "sponza_hard_rot": scenes.sponza_hard_rot, "bistro_proc_rot": scenes.bistro_proc_rot}[args.scene]()
    W, H = args.width, args.height
    n_frames = min(args.steps + args.warmup + (args.verify_frames if world > 1 else 0), args.max_gbuffers)
    common = dict(shadow=True, ao_spp=args.ao_spp, denoise=True, device=local_rank)
    if args.bvh_presplit or args.bvh_frame is not None:
        common["geometry_options"] = {**({"bvh_presplit": args.bvh_presplit} if args.bvh_presplit else {}), **({"bvh_frame": args.bvh_frame} if args.bvh_frame is not None else {})}
    grid = None if args.grid == "auto" else ("strips" if args.grid == "strips" else tuple(int(v) for v in args.grid.lower().split("x")))
    # who moves the halos.  auto = torch.distributed (backend nccl IS RCCL): the route every N > 1 measurement so far has used.  The
    # library's own RCCL calls (--comm c_abi) need one GPU per rank (RCCL refuses two ranks on one device) and have not met a second
    # device yet, so they are never chosen silently: the launcher gives them a separate, time-bounded run (launch_ranks: c_abi_route).
    comm_mode = args.comm if args.comm != "auto" else "torch"
    # (VHR_RCCL_LIBRARY: the library harness.HybridFrameLoop hands to vhr_comm_use_library, for csrc/comm.cpp to load in RCCL's place.  With tests/rccl_shim's stand-in named there, vhr_comm_* runs
    # N ranks on one GPU -- a functional run of the library's own exchange code, which the line then says)
    rccl_override = os.environ.get("VHR_RCCL_LIBRARY") or None
    if comm_mode == "c_abi" and world > 1 and (args.share_device or args.backend != "nccl") and not rccl_override:
        if rank == 0:
            _error_line("--comm c_abi needs one GPU per rank and the nccl backend", args)
        raise SystemExit(2)
    comm_note = None
    if comm_mode == "c_abi" and rccl_override:
        comm_note = f"vhr_comm_* over VHR_RCCL_LIBRARY={os.path.basename(rccl_override)} in RCCL's place: a functional run of the library's exchange code, not a measurement"

    def make_loop(mode):
        return HybridFrameLoop(scene, W, H, n_frames, reflections=_bounces(args), rank=rank, world=world, dist=dist if world > 1 else None,
                               trace_overlap=not args.exchange_raytraced, gather=not args.no_gather, frames_in_flight=args.frames_in_flight,
                               allow_degraded=args.allow_degraded, grid=grid, comm=mode if world > 1 else "torch", **common)
    try:
        loop = make_loop(comm_mode)
    except Exception as e:   # noqa: BLE001  (harness.CommBringUpError is raised on EVERY rank or on none: the ranks stay in step)
        if rank == 0:
            _error_line(f"the frame loop did not come up: {e!r}", args)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)
    ctx = loop.ctx
    if comm_mode == "c_abi" and world > 1:             # the library says which file its RCCL entry points came from (vhr_comm_library)
        comm_note = f"RCCL entry points resolved from {lib.comm_library()}" + (
            " -- handed to vhr_comm_use_library in RCCL's place: a functional run of the library's exchange code, not a measurement" if rccl_override else "")
    build_ms, upload_ms = ctx.build_times_ms()
    k0_builder = "device (binned SAH)" if ctx.bvh_builder_used() == 1 else "host (binned SAH)"
    presplit_level = ctx.bvh_presplit_level()
    _frame = ctx.bvh_frame()
    bvh_frame = "world axes" if np.array_equal(_frame, np.eye(3, dtype=np.float32)) else [[round(float(v), 5) for v in row] for row in _frame]
    option_overrides = {}
    for kv in args.option:
        key, _, val = kv.partition("=")
        ctx.set_option(key, int(val))
        option_overrides[key] = int(val)

    def barrier():
        loop.finish_pending_exchange()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()                          # both streams of a context with frames in flight

    # The frame counter only ever moves forward: the dolly's frame 0 (zero previous matrices, NaN motion) is rendered once, on the
    # clean history it is meant for; beyond the precomputed frames the sequence wraps to frame 1 (HybridFrameLoop.frame_slot).
    f = 0
    strip_check = None
    if world > 1 and args.verify_frames > 0:
        strip_check = verify_strips(args, scene, loop, dist, rank, world, local_rank)
        f = args.verify_frames
        barrier()
        if strip_check != "bit-identical":
            # a decomposition that does not reproduce the single-context frame measures something else: no value is printed
            if rank == 0:
                print(json.dumps({"error": "the ranks' tiles differ from the single-context frame", "strips_vs_single_context": strip_check,
                                  "n_gpus": world}), flush=True)
            loop.close()
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)

    for i in range(f, f + args.warmup):
        loop.frame(i)
    f += args.warmup
    barrier()
    # Only the roofline kernel (a-trous) carries event pairs inside the timed region, and only every 11th / 13th of its launches
    # (a dispatch with an event pair costs ~6 us that the next kernel waits for: all five launches of a frame timed = +30 us on
    # a 0.5 ms frame, every 6th still +5 us; the stride is coprime with the launches per frame, so the sample walks through the
    # step sizes evenly: ~600 samples per second of timed region).  The other kernels are timed in a short extra loop afterwards
    # so that their event records do not sit in the measured frames.
    # "svgf_async_unread" (default at N = 1, one frame in flight): the reference's dead fifth a-trous dispatch leaves the context's stream
    # and runs on the side stream beside the next frame's ray tracing.  The roofline then covers the FOUR launches on the frame's critical
    # path; the side stream's launch is timed as its own kind in the extra loop below and reported beside it.
    async_dead = (world == 1 and args.frames_in_flight == 1 and option_overrides.get("svgf_async_unread", 1) != 0
                  and not option_overrides.get("svgf_elide_unread", 0) and loop.denoise and loop.atrous_steps == 5)
    ATROUS_TIMING_STRIDE = 13 if async_dead else 11       # coprime with the 4 (5) launches per frame on the context's stream: every step size sampled evenly
    ctx.set_option("kernel_timing_stride", ATROUS_TIMING_STRIDE)
    ctx.set_kernel_timing(["svgf_atrous"])
    for k in ("raygen", "svgf_temporal", "svgf_atrous", "svgf_atrous_async", "blit", "reflection"):
        ctx.kernel_time(k, reset=True)
    # ---- the timed region: blocks of exactly --steps frames, repeated until --min-seconds have been measured (a single 20-frame
    # block lasts 14 ms); every block is bracketed by barrier + synchronize, the MEDIAN block is reported ----
    first_timed = f
    def slowest_rank(dt):
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    block_s, f = time_blocks(loop, barrier, f, args.steps, args.min_seconds, slowest=slowest_rank if world > 1 else None)
    ctx.gather_performance_statistics()
    atrous_timed = ctx.kernel_time("svgf_atrous")
    ctx.set_option("kernel_timing_stride", 1)
    ctx.set_kernel_timing(["raygen", "svgf_temporal", "svgf_atrous_async", "blit", "reflection"])
    for i in range(f, f + min(args.steps, 8)):
        loop.frame(i)
    f += min(args.steps, 8)
    barrier()

    rays_per_block = [sum(loop.rays_in_frame(i) for i in range(first_timed + b * args.steps, first_timed + (b + 1) * args.steps)) for b in range(len(block_s))]
    cpu_dev = "cpu" if (world > 1 and args.backend == "gloo") else "cuda"
    t_blocks = torch.tensor(block_s, dtype=torch.float64, device=cpu_dev)
    r_blocks = torch.tensor(rays_per_block, dtype=torch.float64, device=cpu_dev)
    if world > 1:
        dist.all_reduce(t_blocks, op=dist.ReduceOp.MAX)        # a block lasts as long as its slowest rank
        dist.all_reduce(r_blocks, op=dist.ReduceOp.SUM)
    t_blocks, r_blocks = t_blocks.cpu().numpy(), r_blocks.cpu().numpy()
    mid = int(np.argsort(t_blocks)[len(t_blocks) // 2])         # the median block
    dt_max, total_rays = float(t_blocks[mid]), float(r_blocks[mid])

    kt = {k: ctx.kernel_time(k) for k in ("raygen", "svgf_temporal", "svgf_atrous_async", "blit", "reflection")}
    async_dead = async_dead and kt["svgf_atrous_async"][1] > 0
    kt["svgf_atrous"] = atrous_timed
    ctx.set_kernel_timing(False)
    # traversal work counters (one extra, untimed frame with the in-kernel statistics enabled)
    ctx.set_ray_statistics(True)
    loop.frame(f)
    f += 1
    barrier()
    ray_stats, trav_stats = ctx.ray_statistics(), ctx.traversal_statistics()
    ctx.set_ray_statistics(False)
    refl_block = None
    if _bounces(args) and world == 1:
        refl_block = reflection_block(ctx, loop, f, barrier)
        f += 9
    y0, y1 = loop.owned_rows()
    # pixels an a-trous launch computes, averaged over the 5 launches of a frame: strips / tiles shrink the overlap per iteration
    # (tiling.atrous_output_extent: 28, 24, 16, 0, 0 of E = 30); at N = 1 this is W x H
    from vulkanhybridrenderer_amd import tiling

    def computed_pixels(extend):
        cx0, cx1, cy0, cy1 = loop.plan.computed_rect(extend)
        return (cx1 - cx0) * (cy1 - cy0)
    exts = [tiling.atrous_output_extent(loop.plan.overlap, 1 << i) if world > 1 else 0 for i in range(loop.atrous_steps)]
    pixels_svgf = sum(computed_pixels(e) for e in exts) / len(exts)
    atrous_us = kt["svgf_atrous"][0] / max(1, kt["svgf_atrous"][1]) * 1e3
    atrous_bytes = int(ATROUS_BYTES_PER_PIXEL * pixels_svgf)
    pixels_temporal, pixels_owned = computed_pixels(None), computed_pixels(0)
    achieved = atrous_bytes / (atrous_us * 1e-6) / 1e9 if atrous_us > 0 else 0.0
    raygen_ms = kt["raygen"][0] / max(1, kt["raygen"][1])
    passes = {}
    for name in ("Raytrace Pass", "SVGF Denoise Pass"):
        ema, last = ctx.pass_time_ms(name)
        passes[name] = round(last, 4)
    # distribution of the per-pass GPU times (the reference's timestamp pairs, render_graph.cpp:167-199) over 12 more frames,
    # gathered frame by frame outside the timed region: median and p95
    samples = {name: [] for name in passes}
    for i in range(f, f + 12):
        loop.frame(i)
        ctx.gather_performance_statistics()
        for name in samples:
            samples[name].append(ctx.pass_time_ms(name)[1])
    f += 12
    barrier()
    passes_median = {k: round(float(np.median(v)), 4) for k, v in samples.items()}
    passes_p95 = {k: round(float(np.percentile(v, 95)), 4) for k, v in samples.items()}
    rays_one_frame = loop.rays_in_frame(first_timed)
    bvh = ctx.bvh_statistics()
    degraded = list(loop.degraded)
    gather_on, gather_error, plan, tp, rpp, rrpp = loop.gather, loop.gather_error, loop.plan, loop.tp, loop.rays_per_pixel, loop.reference_rays_per_pixel
    trace_overlap = getattr(loop, "trace_overlap", False)
    loop_atrous_steps = loop.atrous_steps
    loop.close()

    # ---- extra blocks of the N = 1 line (each its own context, after the timed region): what raygen.rgen's pass costs with its
    # always-on mirror ray, and the same workload with two frames in flight ----
    extras = {}
    if world == 1 and not args.no_extras:
        def one(options=None, host_k0=None, refl_out=None, **kw):
            lp = HybridFrameLoop(scene, W, H, n_frames, **common, **kw)
            for key, val in {**option_overrides, **(options or {})}.items():
                lp.ctx.set_option(key, val)
            if host_k0 is not None:            # the same scene on the tree the HOST builds ("bvh_builder" 0, csrc/bvh_build.cpp)
                lp.ctx.set_option("bvh_builder", 0)
                lp.ctx.upload_scene(scene)
                host_k0["k0_build_ms"], host_k0["k0_upload_ms"] = (round(v, 1) for v in lp.ctx.build_times_ms())
                host_k0["bvh_max_depth"] = int(lp.ctx.bvh_statistics()["max_depth"])

            def sync():
                torch.cuda.synchronize()
                lp.ctx.synchronize()
            for i in range(args.warmup + 1):
                lp.frame(i)
            first = args.warmup + 1
            ts, nf = time_blocks(lp, sync, first, args.steps, min(args.min_seconds, 0.5))
            b = int(np.argsort(ts)[len(ts) // 2])                        # the median block and its own rays
            rays = sum(lp.rays_in_frame(i) for i in range(first + b * args.steps, first + (b + 1) * args.steps))
            if refl_out is not None:
                refl_out["block"] = reflection_block(lp.ctx, lp, nf, sync)
            lp.close()
            return round(ts[b] / args.steps * 1e3, 4), round(rays / ts[b] / 1e6, 2)
        if not _bounces(args):
            ro = {}
            ms, mr = one(reflections=1, frames_in_flight=args.frames_in_flight, refl_out=ro)
            extras["ms_per_step_with_mirror_ray"] = ms                   # raygen.rgen:59-65 always traces it
            extras["value_with_mirror_ray"] = mr
            extras["traversal_reflection"] = ro.get("block")             # the mirror-ray launch of THAT frame
        # the reference's fifth a-trous iteration is dead work (its output is overwritten before anything reads it); the
        # timed region above executes it like the reference does, this is the same frame with it elided (opt-in "svgf_elide_unread")
        ms, mr = one(options={"svgf_elide_unread": 1}, reflections=_bounces(args), frames_in_flight=args.frames_in_flight)
        extras["ms_per_step_without_dead_iteration"] = ms
        extras["value_without_dead_iteration"] = mr
        # ... and with every dispatch in recorded order on the one stream (option "svgf_async_unread" 0: the dead dispatch back on the
        # frame's critical path instead of on the side stream)
        ms, mr = one(options={"svgf_async_unread": 0}, reflections=_bounces(args), frames_in_flight=args.frames_in_flight)
        extras["ms_per_step_all_dispatches_in_order"] = ms
        extras["value_all_dispatches_in_order"] = mr
        # ... and with svgf.comp run by the ray-tracing kernel in its tiles' epilogues (option "fuse_temporal": one launch and its gap less,
        # the ray-tracing kernel longer by most of what the dispatch cost)
        ms, mr = one(options={"fuse_temporal": 1}, reflections=_bounces(args), frames_in_flight=args.frames_in_flight)
        extras["ms_per_step_temporal_fused_into_ray_tracing"] = ms
        extras["value_temporal_fused_into_ray_tracing"] = mr
        # K0: the timed region above runs on the tree built on the device, where the reference builds its BLAS / TLAS
        # (resource_manager.cpp:650,692,792); this is the same binned-SAH algorithm on the host's cores ("bvh_builder" 0) and the frame on its tree
        hk = {}
        ms, mr = one(host_k0=hk, reflections=_bounces(args), frames_in_flight=args.frames_in_flight)
        hk["ms_per_step"], hk["value"] = ms, mr
        extras["host_k0"] = hk
        other = 2 if args.frames_in_flight == 1 else 1
        ms, mr = one(reflections=_bounces(args), frames_in_flight=other)
        extras[f"ms_per_step_frames_in_flight_{other}"] = ms
        extras[f"value_frames_in_flight_{other}"] = mr

    if rank == 0:
        node_bytes = 32 if option_overrides.get("compact_nodes", 1) else 48
        trav_gbs = (trav_stats["node_visits"] * node_bytes + trav_stats["triangle_tests"] * 48) / max(raygen_ms, 1e-9) / 1e6
        transport = "RCCL (nccl)" if args.backend == "nccl" else "gloo (host memory)"
        from vulkanhybridrenderer_amd import lib as _lib
        fingerprint = _lib.source_fingerprint()
        pmc, pmc_note = load_pmc(args, fingerprint)
        valu = (pmc or {}).get("svgf_atrous_valu_insts_per_launch")
        traffic = (pmc or {}).get("svgf_atrous_mean_traffic_bytes_per_launch")
        traffic = int(traffic) if traffic else None
        # MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues over 2 cycles of its SIMD-32
        valu_floor_us = valu * 2.0 / (SIMDS * CLOCK_HZ) * 1e6 if valu else None
        temporal_us = kt["svgf_temporal"][0] / max(1, kt["svgf_temporal"][1]) * 1e3
        temporal_bytes = int(TEMPORAL_BYTES_PER_PIXEL * pixels_temporal)
        svgf_pass_us = passes_median.get("SVGF Denoise Pass", 0.0) * 1e3
        # (the pass's time stamps cover what the context's stream executes: with the dead dispatch on the side stream, four a-trous launches)
        svgf_pass_bytes = int(temporal_bytes + (loop_atrous_steps - (1 if async_dead else 0)) * atrous_bytes + 3 * 16 * pixels_owned)
        out = {
            "metric": "Mrays/s (unique rays) + ms/frame, Sponza 1080p RT shadows+AO+SVGF",
            "value": round(total_rays / dt_max / 1e6, 2),
            "unit": "Mrays/s",
            # devices the ranks actually ran on: --share-device puts all of them on ONE GPU (a functional check of the tile path over gloo)
            "n_gpus": 1 if args.share_device else world,
            "ranks": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "timed_blocks": {"blocks": len(t_blocks), "steps_per_block": args.steps, "seconds_measured": round(float(t_blocks.sum()), 3),
                             "ms_per_step_min": round(float(t_blocks.min()) / args.steps * 1e3, 4), "ms_per_step_max": round(float(t_blocks.max()) / args.steps * 1e3, 4),
                             "reported": "median block"},
            "config": {
                "workload": f"{scene.name} {W}x{H}: 1 shadow + {args.ao_spp} AO" + ({0: "", 1: " + 1 mirror", 2: " + 1 mirror with a second bounce (second-bounce rays not counted)"}[_bounces(args)]) +
                            " unique rays/px + SVGF (1 temporal + 5 a-trous + 3 blits), 0.05 m/frame dolly",
                "triangles": scene.triangle_count, "primitives": int(len(scene.primitives)),
                "rays_per_covered_pixel": rpp,
                "reference_issued_rays_per_covered_pixel": rrpp,
                "parallelism": (f"row strips x{world}" if plan.grid_cols == 1 else f"screen tiles {plan.grid_rows} x {plan.grid_cols} (rows x columns)") if world > 1 else "single GPU",
                "exchanges_through": None if world == 1 else ("vhr_comm_* (the library's own RCCL calls, csrc/comm.cpp)" if comm_mode == "c_abi" else
                                                              f"torch.distributed P2P (tiling.py), backend {transport}"),
                "exchanges_note": comm_note,
                "frames_in_flight": args.frames_in_flight,
                "schedule": ("reference schedule, every dispatch executed; the fifth a-trous dispatch (output never read, hybrid_render_path.cpp:299-328) is issued on the "
                             "context's side stream beside the next frame's ray tracing (svgf_async_unread, images bit-identical; ms_per_step_all_dispatches_in_order = without)")
                            if async_dead else "reference schedule, every dispatch in recorded order on one stream",
                # every ray of every frame is traced; only the order the launch's workgroups START in comes from the past (their lifetimes two launches ago)
                "ray_tracing_block_order": ("row-major" if option_overrides.get("raygen_cost_order", 1) == 0 else
                                            "longest-lived first (raygen_cost_order: lifetimes two launches ago, sorted by the launch's own first workgroup; images bit-identical; "
                                            "--option raygen_cost_order=0 = row-major)"),
                "options": option_overrides or None,
                "strip_overlap_rows": plan.overlap, "history_halo_rows": plan.halo_rows, "history_halo_cols": plan.halo_cols if plan.grid_cols > 1 else None,
                "overlap_rows_raytraced": ("recomputed locally" if trace_overlap else "exchanged") if world > 1 else None,
                "strips_vs_single_context": strip_check,
                "replan": getattr(loop, "replan_info", None),
                # true only when every rank had a GPU of its own and the bytes moved over RCCL (xGMI between the GPUs of the node)
                "multi_gpu_on_hardware": None if world == 1 else bool(not args.share_device and args.backend == "nccl"),
                "multi_gpu_note": None if world == 1 else (f"{world} ranks on ONE GPU over gloo through host memory: a functional check of the tile path, not a scaling measurement"
                                                           if args.share_device else ("one GPU per rank" + ("" if args.backend == "nccl" else ", exchanges over gloo through host memory"))),
                "final_gather": (f"denoised tiles -> rank 0 every frame (point-to-point over {'vhr_comm_start_frame_exchanges (the same grouped batch as the halos)' if comm_mode == 'c_abi' else transport}, overlapped with the next frame's ray tracing, "
                                 "finished inside the timed region)" if gather_on else ("off" + (f" (disabled at run time: {gather_error})" if gather_error else ""))) if world > 1 else None,
                "degraded": degraded or None,
                "note": "Sponza/lavapipe unavailable (no assets, no Vulkan): procedural stand-in scene; "
                        "raygen.rgen's always-on mirror ray is off unless --reflections (composition discards it in this mode): "
                        "ms_per_step_with_mirror_ray is the frame with it",
            },
            "roofline": {
                "kernel": "svgf_atrous_tile_kernel<step, 4 rows per tile at 1080p (8 at 4K)> (svgf_atrous_filter.comp: LDS comb tiles, weights in the exponent)",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                # the PMC-derived fields (traffic, valu, traversal.address_unit) come from profiles/pmc_<workload>.json and are quoted only when that
                # file was collected on the library loaded here (vhr_source_fingerprint) for this very workload; else null and the reason
                "pmc_file": None if pmc is None else f"profiles/pmc_{workload_key(args)}.json", "pmc_note": pmc_note, "library_fingerprint": fingerprint,
                "traffic_source": "tools/pmc_workload.sh (rocprofv3 --pmc FETCH_SIZE x the correction calibrated in the same run + WRITE_SIZE; includes Infinity-Cache hits; mean over the five "
                                  "launches of a frame.  Three of them also do the work of the pass's three blits -- iteration 0 stores the history copy (+8 B/px) and "
                                  "copies the normals (+16 B/px), iteration 3 stores the Denoised image (+8 B/px): +13 MB on this average and +1.5 / +0.5 us on those "
                                  "launches, which the 24 B/px of `achieved` do not count)",
                # the same mean without the bytes of the three blits those launches carry (32 B/px per frame over five launches, N = 1 only)
                "traffic_less_fused_blit_bytes": (None if traffic is None or world > 1 else int(traffic - 32 * pixels_owned / 5)),
                "avg_launch_us": round(atrous_us, 2), "launches": int(kt["svgf_atrous"][1]),
                "launches_note": (f"HIP event pairs on every {ATROUS_TIMING_STRIDE}th a-trous launch of the timed region on the context's stream "
                                  f"({(4 if async_dead else 5) * args.steps * len(t_blocks)} launches, {'steps 1, 2, 4, 8' if async_dead else 'all five step sizes'} sampled evenly); "
                                  "profiles/*kernel_stats* hold rocprofv3's average over all launches, one row per step size"),
                # the reference's dead fifth dispatch (step 16; nothing reads its output): issued on the side stream, where it fills what the next
                # frame's ray-tracing kernel leaves free -- its launch lasts as long as it shares the chip, which says nothing about the kernel
                # with the mirror ray on, its launch runs beside the SVGF pass (option reflection_async, default): the a-trous launches then share the
                # chip with a kernel of another pass and `achieved` says what they get of it, not what the kernel does alone (the run without
                # --reflections is the kernel's own number)
                "shares_the_chip_with": ("reflection_queue_kernel (the mirror ray's launch on its own stream, reflection_async)"
                                         if _bounces(args) and world == 1 and option_overrides.get("reflection_async", 1) and args.frames_in_flight == 1 else None),
                "side_stream_launch": None if not async_dead else {
                    "what": "svgf_atrous_filter.comp step 16, the dispatch hybrid_render_path.cpp:299-328 never reads (option svgf_async_unread): same kernel, "
                            "same bytes, beside raygen_queue_kernel of the next frame; not on the frame's critical path and not part of `achieved`",
                    "avg_launch_us": round(kt["svgf_atrous_async"][0] / max(1, kt["svgf_atrous_async"][1]) * 1e3, 1), "launches": int(kt["svgf_atrous_async"][1])},
                "algorithmic_bytes_per_launch": int(atrous_bytes),
                # what actually bounds the kernel: issue of its vector instructions (PMC: lanes 96-97 % active, traffic 1.2-1.3 x algorithmic)
                "valu": None if not valu else {
                    "insts_per_launch": valu, "floor_us": round(valu_floor_us, 2), "frac": round(valu_floor_us / atrous_us, 4) if atrous_us > 0 else None,
                    "source": "tools/pmc_workload.sh (rocprofv3 --pmc SQ_INSTS_VALU, wave-level instructions per launch); floor = insts x 2 cycles / (1024 SIMDs x 2.4 GHz), "
                              "a wave64 instruction on a SIMD-32 (MI355X_MICROARCH.md).  The knock-out builds (profiles/r3_atrous_knockouts.txt) price the kernel's mix higher: a packed "
                              "fp32 instruction 4 cycles, a transcendental 8"},
            },
            # the rest of the denoiser against the same HBM peak (VERDICT r2 #2d): svgf.comp alone, and the whole SVGF pass with the
            # reference's schedule (1 temporal + 5 a-trous + 3 blits = 220 B/px) over the pass's median GPU time
            "roofline_temporal": {"kernel": "svgf_temporal_kernel (svgf.comp)", "bound": "hbm", "algorithmic_bytes_per_launch": temporal_bytes,
                                  "avg_launch_us": round(temporal_us, 2), "achieved": round(temporal_bytes / max(temporal_us, 1e-9) / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(temporal_bytes / max(temporal_us, 1e-9) / 1e3 / HBM_PEAK_GBS, 4)},
            "roofline_svgf_pass": {"pass": "SVGF Denoise Pass (hybrid_render_path.cpp:245-331)" + (": the commands on the context's stream (1 temporal + 4 a-trous + 3 blits; the fifth, dead "
                                           "a-trous dispatch runs on the side stream and is outside the pass's time stamps and these bytes)" if async_dead else ""),
                                   "bound": "hbm", "algorithmic_bytes_per_frame": svgf_pass_bytes,
                                   "median_us": round(svgf_pass_us, 1), "achieved": round(svgf_pass_bytes / max(svgf_pass_us, 1e-9) / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(svgf_pass_bytes / max(svgf_pass_us, 1e-9) / 1e3 / HBM_PEAK_GBS, 4)},
            "traversal": {
                "kernel": "raygen_queue_kernel (raygen.rgen's shadow + AO rays + miss.rmiss); the mirror ray runs in reflection_kernel (kernels_us.reflection)",
                "avg_launch_ms": round(raygen_ms, 4),
                "mrays_per_s": round(rays_one_frame * (1 + args.ao_spp) / max(1, rpp) / max(raygen_ms, 1e-9) / 1e3, 1),
                "bvh_nodes": int(bvh["nodes"]), "bvh_bytes": int(bvh["node_bytes"] + bvh["triangle_bytes"]), "bvh_max_depth": int(bvh["max_depth"]),
                "active_lane_utilisation": round(trav_stats["active_lane_utilisation"], 3),
                "node_visits_per_ray": round(trav_stats["node_visits"] / max(1, ray_stats["covered_pixels"] * (1 + args.ao_spp)), 2),
                "triangle_tests_per_ray": round(trav_stats["triangle_tests"] / max(1, ray_stats["covered_pixels"] * (1 + args.ao_spp)), 2),
                # bytes the walk asks for: a 32-byte half-precision node per visit (48-byte nodes with compact_nodes 0), a 48-byte record per triangle test
                "effective_traversal_gbs": round(trav_gbs, 1),
                # against the level the cache-resident tree is read from (L2 ~34.5 TB/s), not HBM
                "l2_frac": round(trav_gbs / L2_PEAK_GBS, 4),
                "stack_overflows": int(ray_stats["stack_overflows"]),
                "address_unit": address_unit_block(pmc, raygen_ms),
                "note": "counters cover the any-hit (shadow + AO) queue kernel; utilisation = (node visits + triangle tests) / (64 x wave-level trips of those loops)",
            },
            # K0: the reference builds its BLAS / TLAS on the device once per scene (resource_manager.cpp:650,692,792); so does this
            # (binned SAH, csrc/kernels_bvh.hip), once per scene, outside the frame.  upload = scene arrays + the tree fetched back for the host's self-checks
            "k0_build_ms": round(build_ms, 1), "k0_upload_ms": round(upload_ms, 1), "k0_builder": k0_builder, "bvh_frame": bvh_frame, "bvh_presplit": {"percent": args.bvh_presplit, "level": presplit_level} if args.bvh_presplit else None,
            "kernels_us": {"svgf_temporal": round(kt["svgf_temporal"][0] / max(1, kt["svgf_temporal"][1]) * 1e3, 2),
                           "svgf_atrous": round(atrous_us, 2),
                           "blit": round(kt["blit"][0] / max(1, kt["blit"][1]) * 1e3, 2) if kt["blit"][1] else None,      # None: all three blits are stores of a-trous launches
                           "reflection": round(kt["reflection"][0] / max(1, kt["reflection"][1]) * 1e3, 2) if kt["reflection"][1] else None},
            "passes_ms": passes, "passes_ms_median": passes_median, "passes_ms_p95": passes_p95,
        }
        if refl_block is not None:
            out["traversal_reflection"] = refl_block
        out.update(extras)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, tp, args.cpu_frames, rpp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
