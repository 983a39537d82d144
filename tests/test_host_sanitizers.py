"""Address / undefined-behaviour sanitizer build of the library's host-only translation unit.

``scasml_gp_amd/csrc/plan_host.cpp`` holds every entry point that runs on the host alone -- ABI bookkeeping, the schedule helpers
(site kinds, dealing of the Monte-Carlo units) and the normal table -- in plain C++ with no HIP header, so the SAME file that
hipcc compiles into libscasml_hip.so is built here with ``g++ -fsanitize=address,undefined`` and driven with exactly-sized heap
buffers (tests/host/sanitize_driver.cpp).  GPU AddressSanitizer is not available on the pool; the device code is covered by
the parity tests instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sanitizer_toolchain(tmp_path):
    """Skip (not fail) on a machine without g++ or its sanitizer runtimes: the suite's subject is the library, not the host's toolchain."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not installed")
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    r = subprocess.run(["g++", "-fsanitize=address,undefined", str(probe), "-o", str(tmp_path / "probe")], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("g++ cannot link -fsanitize=address,undefined here: " + r.stderr[-200:])


def test_host_entry_points_under_asan_and_ubsan(tmp_path):
    _sanitizer_toolchain(tmp_path)
    exe = str(tmp_path / "sanitize_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "scasml_gp_amd", "csrc"),
           os.path.join(ROOT, "scasml_gp_amd", "csrc", "plan_host.cpp"), os.path.join(ROOT, "tests", "host", "sanitize_driver.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "host sanitizer driver ok" in run.stdout


def test_the_sanitizer_build_catches_the_abi2_overrun(tmp_path):
    """The class of bug behind round 2's unexplained abort (profiles/HISTORY.md, round-4 section 8): ABI 2's scasml_normal_table(float*) copied the library's
    whole table into the caller's buffer.  The same copy into a buffer one row short, under the sanitizer: it must be reported --
    this is what says the build above would have caught it."""
    _sanitizer_toolchain(tmp_path)
    src = tmp_path / "overrun.cpp"
    src.write_text('#include <string.h>\n#include <vector>\nstatic const float T[768][4] = {{1.0f}};\n'
                   'int main() { std::vector<float> b(767 * 4); memcpy(b.data(), T, sizeof(T)); return b[0] == 1.0f ? 0 : 1; }\n')
    exe = str(tmp_path / "overrun")
    assert subprocess.run(["g++", "-O1", "-g", "-fsanitize=address", str(src), "-o", exe], capture_output=True).returncode == 0
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert run.returncode != 0 and "heap-buffer-overflow" in run.stderr
