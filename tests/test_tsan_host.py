"""CPU-only: ThreadSanitizer run of libmvn_hip's HOST code (no GPU involved: the GPU pool refuses sanitizer builds, so this
file is listed in .gpurunignore and never travels to a GPU box; it runs with the `-m "not gpu"` tests in the build container)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SANITIZE = "-fsanitize=thread"  # host code only; this file is in .gpurunignore (never runs on a GPU box)


def test_host_threads_under_tsan():
    """The library's process-global state (MVN_* switch table, dynamic-LDS opt-in table, CU-count cache) is safe to use from
    several host threads: csrc/mvn_hip.hip's HOST code is built with the host ThreadSanitizer together with a driver that calls
    argument-validating and dispatch-query entry points from eight threads while one re-reads the switches
    (tests/native/tsan_driver.cpp).  No device needed, none used."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "tsan_driver")
        build = subprocess.run([hipcc, "--offload-arch=gfx950", "--offload-host-only", SANITIZE, "-O1", "-g", "-std=c++17",
                                "-ffp-contract=off", "-x", "hip", os.path.join(ROOT, "meta-viterbinet_amd", "csrc", "mvn_hip.hip"),
                                "-x", "c++", os.path.join(ROOT, "tests", "native", "tsan_driver.cpp"), "-o", exe, "-pthread",
                                # host-only: the device code object the registration stub points at does not exist (never used)
                                "-Wl,--unresolved-symbols=ignore-all"], capture_output=True, text=True)
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=0"))
    assert "ThreadSanitizer" not in run.stderr, run.stderr[:3000]
    assert run.returncode == 0 and "0 wrong answers" in run.stdout, (run.returncode, run.stdout, run.stderr[-500:])
