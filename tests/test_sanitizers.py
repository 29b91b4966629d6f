"""CPU sanitizer runs (SURVEY.md section 5 "race detection / sanitizers": the reference has the Vulkan validation layer only).  The host
side of the library -- pass registry and execution order (render_graph.cpp), the host BVH builder and its node forms (bvh_build.cpp), the
strip / tile planners (comm.cpp), both re-hosted render paths, the SVGF state blob -- and the oracle are rebuilt under AddressSanitizer +
UndefinedBehaviorSanitizer and the host-only tests are run on those builds in a child process with the sanitizer runtime preloaded
(python itself is not instrumented).  CPU only: GPU AddressSanitizer is not available on the GPU pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vulkanhybridrenderer_amd", "csrc")


def _run(preload, env_extra, args):
    env = dict(os.environ)
    env.update(env_extra)
    env["LD_PRELOAD"] = preload
    # leaks: CPython keeps allocations alive at exit by design; the rest stops the child at the first report
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["OMP_NUM_THREADS"] = "4"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out, out[-2000:]
    return out


def test_host_side_of_the_library_under_asan_and_ubsan():
    import shutil
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc is not installed: the instrumented build of the library's host side needs it")
    subprocess.run(["make", "-C", CSRC, "-s", "-j8", "asan"], check=True)
    rt = subprocess.run(["make", "-C", CSRC, "-s", "asan-runtime"], check=True, capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("the compiler's AddressSanitizer runtime is not installed")
    so = os.path.join(ROOT, "vulkanhybridrenderer_amd", "libvhr_amd_asan.so")
    out = _run(rt, {"VHR_TEST_LIB": so},
               ["tests/test_graph_host.py", "tests/test_abi.py", "tests/test_comm_plan.py", "tests/test_raytraced_path.py", "tests/test_screen_space.py"])
    # the run really was on the instrumented library
    maps = subprocess.run([sys.executable, "-c",
                           "import os; from vulkanhybridrenderer_amd import lib; lib.LIB_PATH = os.environ['VHR_TEST_LIB']; lib.load(); "
                           "print([l.split()[-1] for l in open('/proc/self/maps') if 'libvhr_amd' in l][0])"],
                          cwd=ROOT, env=dict(os.environ, LD_PRELOAD=rt, VHR_TEST_LIB=so, ASAN_OPTIONS="detect_leaks=0"), capture_output=True, text=True)
    assert maps.stdout.strip().endswith("libvhr_amd_asan.so"), maps.stdout + maps.stderr
    assert out


def test_oracle_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True)
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc's libasan.so is not installed")
    _run(rt, {"VHR_ORACLE_LIB": os.path.join(ROOT, "oracle", "libvhr_oracle_asan.so")}, ["tests/test_oracle_kat.py"])
