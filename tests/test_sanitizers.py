"""CPU only: the oracle's C restatement and the HOST-ONLY slice of the product library
(sdrx_fifo_*, sdrx_sdriq_*, sdrx_chan_plan) under AddressSanitizer + UBSan.  `make -C oracle asan` builds both;
the golden suites run through the oracle build, tests/host_asan_driver.py through the host slice.  Never on the GPU
box: no kernel is launched here, and GPU sanitizers are not available on the pool."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "oracle", "_asan")


def _run(preload, extra_env, argv):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **extra_env)
    p = subprocess.run(argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    return out


@pytest.fixture(scope="module")
def asan_build():
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("compilers of the build container absent")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)


def test_oracle_golden_suites_under_asan_ubsan(asan_build):
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt):
        pytest.skip("gcc has no libasan")
    out = _run(rt, {"SDRO_LIB": os.path.join(ASAN, "libsdro_asan.so")},
               [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                "tests/test_oracle_golden.py", "tests/test_oracle_float_golden.py", "tests/test_oracle_wide24_golden.py"])
    assert " passed" in out


def test_product_host_slice_under_asan_ubsan(asan_build):
    rts = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rts:
        pytest.skip("clang's sanitizer runtime absent")
    out = _run(rts[-1], {}, [sys.executable, "tests/host_asan_driver.py", os.path.join(ASAN, "libsdrx_host_asan.so")])
    assert "ok" in out
