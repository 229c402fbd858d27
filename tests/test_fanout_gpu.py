"""GPU: sdrx_fanout_* -- one staged stream copied to several destinations by peer copies.  On a one-GPU box both destinations are
the source device itself (the call degenerates to device-to-device copies): that checks the plumbing (events, streams, buffers,
ordering against a producer stream); on a multi-GPU box the destinations are spread over the other GPUs and the copies run over xGMI."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc
from tests import synth

pytestmark = pytest.mark.gpu


def test_fanout_feeds_identical_banks():
    torch = pytest.importorskip("torch")
    n_dev = torch.cuda.device_count()
    dsts = [(1 + i) % n_dev for i in range(2)] if n_dev > 1 else [0, 0]
    n = 300_000
    x = synth.mix(n, 9, 2047, 700)
    src = torch.from_numpy(x).to("cuda:0")
    torch.cuda.synchronize()
    f = sa.Fanout(0, dsts, 4 * n)
    f.send(src.data_ptr(), 4 * n)
    rates, fcs = [48000, 12500], [100_000, -555_000]
    want = [orc.Chain(orc.chan_plan(2_400_000, r, c)[0]).feed(x) for r, c in zip(rates, fcs)]
    for i, dev in enumerate(dsts):
        f.wait(i)
        bank = sa.ChannelizerBank(2_400_000, rates, fcs, device=dev)
        bank.feed_dev(f.buffer(i), n)
        for c in range(2):
            assert np.array_equal(bank.read(c), want[c]), (i, dev, c)
        bank.close()
    # a second, shorter send reuses the buffers; too many bytes are refused
    f.send(src.data_ptr(), 4 * 1000)
    f.wait(0); f.wait(1)
    assert sa.lib().sdrx_fanout_send(f._h, src.data_ptr(), 4 * n + 4, None) == -1
    assert sa.lib().sdrx_fanout_buffer(f._h, 5) in (None, 0)
    f.close()
