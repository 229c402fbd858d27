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
