"""bench.py --gpus N (N > 1) without WORLD_SIZE must start N ranks itself (a child `python -m torch.distributed.run`), never measure
one GPU under an N-GPU label, and fail loudly when the devices are not there.  No GPU involved here: the launcher
itself makes no HIP call, and this container has no device, which is exactly the error path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r


def test_dry_launch_prints_the_torchrun_command():
    rc, line, r = run(["--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-launch"])
    assert rc == 0, r.stderr[-2000:]
    cmd = line["dry_launch"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]           # the same arguments, minus --dry-launch
    assert line["n_gpus"] == 4
    import torch
    assert line["devices_visible"] == torch.cuda.device_count()
    assert line["would_run"] == (torch.cuda.device_count() >= 4)


def test_missing_devices_is_an_error_not_a_one_gpu_run():
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this machine has the devices")
    rc, line, r = run(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert rc == 2
    assert line is not None and "error" in line and line["n_gpus"] == 8 and line["value"] is None
    assert line["devices_visible"] == torch.cuda.device_count()
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1         # one line, and it is the error


def test_under_torchrun_a_rank_without_a_device_is_an_error_too():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this machine has the devices")
    rc, line, r = run(["--gpus", "2"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2
    assert "error" in line and line["n_gpus"] == 2
    rc, line, r = run(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})      # --gpus and WORLD_SIZE disagree
    assert rc == 2 and "WORLD_SIZE" in line["error"]


def test_one_rank_needs_no_launcher():
    rc, line, r = run(["--gpus", "1", "--dry-launch"])
    assert rc == 0 and line["dry_launch"] is None


def test_ranks_stop_the_timed_region_together(tmp_path):
    """bench.time_blocks with two gloo ranks: the stop decision is taken on the slowest rank's time of each block, so both ranks run
    the same count of blocks whatever their own clocks say (r4: a rank that had measured --min-seconds on its own clock left the timed
    region one block early and the other rank waited in a barrier for ever)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for attempt in range(3):                         # (the threshold sits at a block boundary: several tries, every one must agree)
        out = tmp_path / f"counts{attempt}.txt"
        port = 29871 + attempt
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "tests", "time_blocks_worker.py"), str(out), "1"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        a, b = out.read_text().split()
        assert a == b and int(a) >= 1


def test_counter_files_are_quoted_only_for_the_library_and_workload_they_were_collected_on(tmp_path, monkeypatch):
    """bench.load_pmc: profiles/pmc_<workload>.json is quoted when its fingerprint equals the loaded library's and its
    workload is the one being measured at N = 1; a stale fingerprint, another workload or N > 1 give (None, the reason)."""
    import argparse
    import bench
    args = argparse.Namespace(gpus=1, gltf=None, scene="sponza_proc", width=1920, height=1080, ao_spp=2, reflections=False, refl_bounces=0)
    key = bench.workload_key(args)
    assert key == "sponza_proc_1920x1080_ao2_refl0"
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / f"pmc_{key}.json").write_text(json.dumps({"fingerprint": "abc", "workload": key, "svgf_atrous_mean_traffic_bytes_per_launch": 1.0}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    d, why = bench.load_pmc(args, "abc")
    assert d is not None and why is None and d["workload"] == key
    d, why = bench.load_pmc(args, "def")
    assert d is None and "stale" in why and "abc" in why and "def" in why
    args.ao_spp = 4
    d, why = bench.load_pmc(args, "abc")
    assert d is None and "no profiles/pmc_sponza_proc_1920x1080_ao4_refl0.json" in why
    args.ao_spp, args.gpus = 2, 2
    d, why = bench.load_pmc(args, "abc")
    assert d is None and "N > 1" in why


def test_committed_counter_files_name_their_workload_and_one_library():
    """The counter files under profiles/ (one per BASELINE configuration measured on one GPU) carry their workload key in the file name
    and inside, the fields bench.py reads, and ONE library fingerprint between them (they are collected together, on the round's
    final library; bench.py refuses them on any other)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "pmc_*.json")))
    assert len(files) >= 4
    prints = set()
    for f in files:
        d = json.load(open(f))
        assert os.path.basename(f) == f"pmc_{d['workload']}.json"
        assert d["svgf_atrous_mean_traffic_bytes_per_launch"] > 0 and d["svgf_atrous_valu_insts_per_launch"] > 0
        ta = d["raygen_ta"]
        assert "raygen_queue_kernel" in ta["kernel"] and 0.0 < ta["ta_busy_frac"] < 1.0 and 10.0 < ta["ta_cycles_per_load_instruction"] < 64.0
        prints.add(d["fingerprint"])
    assert len(prints) == 1


def test_help_text_formats():
    """`python bench.py --help` prints the usage (argparse formats every help string with %: a bare per-cent sign in one of them used to abort it)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--bvh-frame" in out.stdout and "--gpus" in out.stdout, out.stderr[-500:]
