"""GPU: the matrix-core half-band primitive on its own (sdrangel_amd/csrc/hb_mfma.hpp, v_mfma_i32_16x16x64_i8 on packed
int16 arms).  tools/ubench_hb_i8 (built by __graft_entry__.build()) feeds it full-range int16 data incl. runs of -32768 /
32767 and compares every output of a tile with a host loop in 64-bit arithmetic: orders 48 (DownChannelizer stages,
inthalfbandfiltereo.h:792-830) and 64 (Decimators, :832-870), plain and alternating-sign taps.  Bit-exact or fail."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_i8_mfma_halfband_primitive_is_exact_on_the_device():
    exe = os.path.join(ROOT, "tools", "ubench_hb_i8")
    if not os.path.exists(exe):
        pytest.fail("tools/ubench_hb_i8 not built: run __graft_entry__.build()")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if "mismatches" in ln]
    assert len(lines) == 4 and all(ln.rstrip().endswith(" 0 mismatches") for ln in lines), out.stdout
    assert "ALL EXACT" in out.stdout
