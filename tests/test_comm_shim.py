"""The library's own multi-GPU exchanges (vhr_comm_*, csrc/comm.cpp: plan -> pieces -> pack -> one grouped batch of ncclSend / ncclRecv ->
unpack, the gather, the `broken` state) at world size 2 and 4 on ONE GPU.  RCCL refuses two ranks per device, so the eight RCCL entry
points comm.cpp resolves come from tests/rccl_shim (a stand-in over /dev/shm files; VHR_RCCL_LIBRARY tells bench.py / the worker, which hand it to vhr_comm_use_library): every byte of the
N > 1 path moves through the product's code -- only the wire is replaced.  Checked: every rank's Denoised and Reflections tile and the
frame gathered on rank 0 equal the single context's bit for bit (bench.py's own pre-timing verification), the same through
torch.distributed (tiling.StripExchanges over gloo); one injected failure per error path of a batch -- every rank ends by itself
with a non-zero code, none hangs.  NOT checked by this route: the stand-in synchronises its streams on the host and copies with blocking calls at
ncclGroupEnd, so a missing stream or event dependency in comm.cpp (its own stream against the context's, the reuse of its staging buffers)
cannot show here -- the order of those dependencies against RCCL's asynchronous transport waits for a box with two devices."""
import os
import subprocess
import sys
import time

import pytest

from tests.test_comm_plan import _bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "rccl_shim")
SHIM = os.path.join(SHIM_DIR, "librccl_shim.so")


@pytest.fixture(scope="module")
def shim():
    subprocess.run(["make", "-C", SHIM_DIR, "-s"], check=True)
    assert os.path.exists(SHIM)
    return SHIM


@pytest.mark.parametrize("world,extra", [(2, ["--grid", "strips"]), (2, ["--grid", "1x2", "--exchange-raytraced"]), (3, ["--grid", "1x3"]), (3, ["--grid", "strips", "--exchange-raytraced"]),
                                         (4, ["--grid", "2x2"]), (4, ["--grid", "auto", "--refl-bounces", "2"])])
def test_c_abi_exchanges_on_one_gpu_equal_the_single_context_and_the_torch_route(shim, world, extra, monkeypatch):
    args = ["--gpus", str(world), "--share-device", "--scene", "tiny", "--width", "320", "--height", "200", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05",
            "--no-cpu-baseline", "--no-extras", "--verify-frames", "3", "--reflections"] + extra
    monkeypatch.setenv("VHR_RCCL_LIBRARY", shim)
    monkeypatch.setenv("VHR_RCCL_SHIM_TAG", _TAG)
    monkeypatch.setenv("VHR_RCCL_SHIM_TIMEOUT_S", "60")
    r, line = _bench(args + ["--comm", "c_abi"], timeout=600)
    assert r.returncode == 0 and line and "error" not in line, (r.stdout[-2000:], r.stderr[-3000:])
    assert line["ranks"] == world and line["n_gpus"] == 1
    assert "vhr_comm" in line["config"]["exchanges_through"] and "librccl_shim" in line["config"]["exchanges_note"]
    assert line["config"]["strips_vs_single_context"] == "bit-identical"          # Denoised + Reflections tiles of every rank + the gathered frame
    assert "finished inside the timed region" in line["config"]["final_gather"]
    monkeypatch.delenv("VHR_RCCL_LIBRARY")
    r, line2 = _bench(args + ["--comm", "torch"], timeout=600)
    assert r.returncode == 0 and line2["config"]["strips_vs_single_context"] == "bit-identical", (r.stdout[-2000:], r.stderr[-3000:])
    assert line2["config"]["parallelism"] == line["config"]["parallelism"]
    _sweep_shim_directories()


def test_c_abi_replan_between_frames_on_one_gpu(shim, monkeypatch):
    """vhr_comm_replan (round 6) at world 4 on one GPU through the stand-in: after verified frame 1 rank 0 is declared twice as slow, every rank cuts the grid again
    from the same refined map, the communicator and the context take the new plan, the temporal history, the moments history and the previous normals follow their
    pixels through the library's own plan -> pieces -> pack -> grouped batch -> unpack code -- and verified frames 2-3 on the NEW rectangles, and the frame
    gathered on rank 0, equal the single context's bit for bit."""
    args = ["--gpus", "4", "--share-device", "--width", "640", "--height", "360", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05",
            "--no-cpu-baseline", "--no-extras", "--verify-frames", "4", "--replan-frame", "1", "--reflections", "--grid", "2x2", "--comm", "c_abi"]
    monkeypatch.setenv("VHR_RCCL_LIBRARY", shim)
    monkeypatch.setenv("VHR_RCCL_SHIM_TAG", _TAG)
    monkeypatch.setenv("VHR_RCCL_SHIM_TIMEOUT_S", "60")
    monkeypatch.setenv("VHR_BENCH_REPLAN_TIMES", "2,1,1,1")
    r, line = _bench(args, timeout=600)
    assert r.returncode == 0 and line and "error" not in line, (r.stdout[-2000:], r.stderr[-3000:])
    assert "vhr_comm" in line["config"]["exchanges_through"] and line["config"]["strips_vs_single_context"] == "bit-identical"
    rp = line["config"]["replan"]
    assert rp["after_frame"] == 1 and rp["rect_before"] != rp["rect_after"]
    _sweep_shim_directories()


def _run_ranks(shim, world, frames, fail, fail_rank, mode="", timeout_s=8):
    port = str(_free_port())
    env = dict(os.environ, VHR_RCCL_LIBRARY=shim, VHR_RCCL_SHIM_TAG=_TAG, VHR_RCCL_SHIM_TIMEOUT_S=str(timeout_s), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VHR_RCCL_SHIM_FAIL", None)
    if fail:
        env.update(VHR_RCCL_SHIM_FAIL=fail, VHR_RCCL_SHIM_FAIL_RANK=str(fail_rank))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_shim_worker.py"), str(r), str(world), str(frames), port, mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs, codes, hung = [], [], False
    deadline = time.time() + 240
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            hung = True
            p.kill()                                   # exactly the process started above
            out, _ = p.communicate()
        outs.append(out)
        codes.append(p.returncode)
    _sweep_shim_directories()
    return codes, outs, hung


_TAG = "t%d%x" % (os.getpid(), int(time.time() * 1e3) & 0xffffff)      # this module run's mark in the shim's directory names (VHR_RCCL_SHIM_TAG)


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _sweep_shim_directories():
    """A run that ends in an injected failure leaves its message directory behind (the shim removes it when the last rank destroys its
    communicator): /dev/shm is memory, so the test cleans up what the shim created FOR THIS MODULE RUN (its tag in the name) -- not another run's."""
    import glob
    import shutil
    for d in glob.glob(f"/dev/shm/vhr_rccl_shim_{_TAG}_*"):
        shutil.rmtree(d, ignore_errors=True)


def test_shim_ranks_without_a_launcher_run_clean(shim):
    codes, outs, hung = _run_ranks(shim, 2, 4, None, 0)
    assert not hung and codes == [0, 0], outs


@pytest.mark.parametrize("world,fail,fail_rank,mode", [(2, "send:2", 1, ""), (2, "recv:1", 0, ""), (2, "groupstart:2", 0, ""), (2, "groupend:1", 1, ""),
                                                        (4, "send:5", 2, ""), (2, "send:1", 0, "exchange_raytraced")])
def test_an_injected_failure_ends_every_rank_and_hangs_none(shim, world, fail, fail_rank, mode):
    """One failing call inside a frame's batch (ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd; exchange #1's batch on the context's
    stream too): the failing rank's vhr_comm_* call returns the error with the group closed, its communicator refuses further starts and
    still drains; the peers' receives fail instead of waiting for ever; every rank exits non-zero by itself."""
    codes, outs, hung = _run_ranks(shim, world, 6, fail, fail_rank, mode)
    text = "\n".join(outs)
    assert not hung, text[-3000:]
    assert all(c == 3 for c in codes), (codes, text[-3000:])
    assert f"rank {fail_rank}: exchange failed" in text and "injected failure" in text
    assert f"rank {fail_rank}: start after the failure refused" in text and "unusable" in text


def test_a_communicator_that_fails_to_come_up_is_refused_on_every_rank(shim):
    """ncclCommInitRank fails on one rank: harness.HybridFrameLoop's collective bring-up raises CommBringUpError on EVERY rank (exit 4)."""
    codes, outs, hung = _run_ranks(shim, 2, 3, "init:1", 1)
    assert not hung and codes == [4, 4], (codes, "\n".join(outs)[-3000:])


def test_a_forced_rccl_library_that_does_not_load_is_an_error(monkeypatch):
    """vhr_comm_use_library names the library to use and no other: a path that does not load fails vhr_comm_get_unique_id, it does not fall through to the
    installation's RCCL; the library itself reads NO environment variable for this (round 6: VHR_RCCL_LIBRARY is honoured by the host tooling's explicit call,
    lib.comm_use_library_from_environment, nowhere else); vhr_comm_library names the file the entry points came from; once they are bound the choice is closed.
    (Child processes: the loader's choice is made once per process.)"""
    head = "import sys; sys.path.insert(0, %r)\nfrom vulkanhybridrenderer_amd import lib\n" % ROOT
    tail = "try:\n    lib.Comm.unique_id()\n    print('LOADED', lib.comm_library())\nexcept lib.VhrError as e:\n    print('REFUSED', lib.comm_library())\n"
    run = lambda code, env: subprocess.run([sys.executable, "-c", head + code + tail], env=env, capture_output=True, text=True, timeout=300)   # noqa: E731
    bad = dict(os.environ, VHR_RCCL_LIBRARY="/nonexistent/librccl.so")
    r = run("lib.comm_use_library('/nonexistent/librccl.so')\n", dict(os.environ))
    assert "REFUSED" in r.stdout and "LOADED" not in r.stdout and "vhr_comm_use_library" in r.stdout, r.stdout + r.stderr
    r = run("", bad)                                     # the variable alone steers nothing
    assert ("LOADED" in r.stdout and "nonexistent" not in r.stdout) or "RCCL not found" in r.stdout, r.stdout + r.stderr
    r = run("lib.comm_use_library_from_environment()\n", bad)      # ... until the host program passes it on
    assert "REFUSED" in r.stdout and "LOADED" not in r.stdout, r.stdout + r.stderr
    r = run("lib.comm_library()\ntry:\n    lib.comm_use_library('/tmp/x.so')\n    print('ACCEPTED LATE')\nexcept lib.VhrError as e:\n    print('TOO LATE')\n", dict(os.environ))
    assert "TOO LATE" in r.stdout and "ACCEPTED LATE" not in r.stdout, r.stdout + r.stderr


def test_the_stand_in_is_named_by_the_library(shim):
    """vhr_comm_library reports the stand-in when the host handed it over."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom vulkanhybridrenderer_amd import lib\nlib.comm_use_library(%r)\nlib.Comm.unique_id()\nprint('FROM', lib.comm_library())\n" % (ROOT, shim))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VHR_RCCL_SHIM_TIMEOUT_S="5"), capture_output=True, text=True, timeout=300)
    assert "FROM" in r.stdout and "librccl_shim" in r.stdout, r.stdout + r.stderr
