"""experiments/mfma_decim (NOT the product path): decimate64_cen <16,12> with every half-band FIR on
v_mfma_f32_16x16x32_f16.  Bit-exactness against the oracle and against the product
kernel; the rate is measured by tools/mfma_experiment_rate.py.  Skipped when the experiment library is not built."""
import ctypes as C
import os

import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "experiments", "mfma_decim", "libmfx.so")


def mfx():
    if not os.path.exists(SO):
        pytest.skip("experiments/mfma_decim/libmfx.so not built")
    sa.lib()                                                   # one HIP runtime (see sdrangel_amd/__init__.py)
    L = C.CDLL(SO)
    L.mfx_decim64.restype = C.c_int
    L.mfx_decim64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    return L


def run(L, x, hist=None, spw=32):
    n = x.size // 2
    d_in = torch.from_numpy(x).cuda()
    d_hist = torch.zeros(2 * 4096, dtype=torch.int16, device="cuda") if hist is None else torch.from_numpy(hist).cuda()
    d_out = torch.zeros(2 * (n // 64) + 64, dtype=torch.int16, device="cuda")
    d_flags = torch.zeros((n + 4095) // 4096 + 1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = L.mfx_decim64(d_hist.data_ptr(), d_in.data_ptr(), d_out.data_ptr(), d_flags.data_ptr(), n, spw, None)
    assert rc == 0
    torch.cuda.synchronize()
    return d_out[: 2 * (n // 64)].cpu().numpy(), d_flags.cpu().numpy()


@pytest.mark.parametrize("n,spw,amp,tone", [(64 * 1024, 32, 2047, (0.0021, 1000)), (1 << 20, 32, 2047, None), (3 * 4096 + 64 * 5, 4, 1500, (0.11, 500)),
                                            (1 << 22, 64, 2047, (0.0005, 2047))])
def test_mfma_chain_is_bit_exact(n, spw, amp, tone):
    L = mfx()
    x = orc.synth_iq(n, seed=7 + n % 97, amp=amp, tone=tone)
    x = np.clip(x, -2048, 2047).astype(np.int16)               # the <16,12> contract
    got, flags = run(L, x, spw=spw)
    want = orc.Decim(6, sa.FC_CEN, 12).process(x)
    assert flags.sum() == 0
    assert got.size == want.size
    assert np.array_equal(got, want), int((got != want).sum())


def test_mfma_chain_extremes_of_the_contract_and_state_carry():
    L = mfx()
    n = 1 << 18
    x = np.empty(2 * n, np.int16)
    x[0::2] = np.where(np.arange(n) % 2 == 0, 2047, -2048)     # worst-case alternating full scale on I
    x[1::2] = -2048
    x[: 2 * 9000] = orc.synth_iq(9000, seed=3, amp=2047)
    o = orc.Decim(6, sa.FC_CEN, 12)
    cut = 2 * 65536
    w1, w2 = o.process(x[:cut]), o.process(x[cut:])
    g1, f1 = run(L, x[:cut])
    g2, f2 = run(L, x[cut:], hist=x[cut - 2 * 4096: cut])      # carried state = the last 4096 input samples
    assert f1.sum() == 0 and f2.sum() == 0
    assert np.array_equal(g1, w1) and np.array_equal(g2, w2)


def test_mfma_chain_flags_input_outside_the_contract():
    L = mfx()
    x = orc.synth_iq(1 << 16, seed=5, amp=2047)
    x[2 * 20000 + 1] = 2048                                     # one int16 just outside [-2048, 2047]
    _, flags = run(L, x)
    assert flags[20000 // 4096] == 1 and flags[: 20000 // 4096].sum() == 0
