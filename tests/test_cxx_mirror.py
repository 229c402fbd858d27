"""The C++ mirror header (include/sdrx/dsp.hpp) compiles against the C ABI and runs: host-only checks without a
GPU, one decimate64_cen + a 2-channel bank through the reference-named classes when a GPU is present."""
import os
import subprocess
import tempfile

import sdrangel_amd as sa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_mirror_builds_and_runs():
    exe = os.path.join(tempfile.mkdtemp(), "cxx_mirror_check")
    libdir = os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx_mirror_check.cpp"), "-o", exe,
                           "-L" + libdir, "-lsdrx", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "cxx mirror" in out.stdout
