import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference compiled into oracle/_ref (build container only)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle) once per session; the product .so is built by __graft_entry__.build()."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libsdro.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("sdro.c", "sdro_float.c", "sdro_fdecim.c", "sdro_audio.c", "sdro.h")]
    src = [s for s in src if os.path.exists(s)]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsdro.so"])
    yield


# The half-band engines: the matrix-core kernels are the default; the dot2 kernels of rounds
# 1-2 stay as the second gfx950 implementation.  Every GPU test of the integer decimators and of the channelizer bank
# runs under BOTH (the libraries read the switch when a handle / a bank plan is created).
_ENGINE_MODULES = {
    "test_decim_gpu": "SDRX_DECIM_ENGINE", "test_golden_gpu": "SDRX_DECIM_ENGINE", "test_fullsize_gpu": "SDRX_DECIM_ENGINE",
    "test_decim_switch_gpu": "SDRX_DECIM_ENGINE", "test_random_splits_gpu": "SDRX_DECIM_ENGINE", "test_decim_batch_gpu": "SDRX_DECIM_ENGINE",
    "test_chan_gpu": "SDRX_CHAN_ENGINE", "test_wide_banks_gpu": "SDRX_CHAN_ENGINE", "test_bank_fuzz_gpu": "SDRX_CHAN_ENGINE",
}


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in _ENGINE_MODULES and "hb_engine" in metafunc.fixturenames:
        metafunc.parametrize("hb_engine", ["mfma", "valu"], indirect=True)


@pytest.fixture(autouse=True)
def hb_engine(request, monkeypatch):
    mod = request.module.__name__.rsplit(".", 1)[-1]
    eng = getattr(request, "param", None)
    if mod in _ENGINE_MODULES and eng is not None:
        # test_golden_gpu / test_fullsize_gpu hold decimator AND bank cases: both switches follow the parameter
        monkeypatch.setenv("SDRX_DECIM_ENGINE", eng)
        monkeypatch.setenv("SDRX_CHAN_ENGINE", eng)
    yield eng
