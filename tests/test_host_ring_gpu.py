"""Pinned, double-buffered host path of the decimator (sdrx_decim_ring_*): the device thread's receive buffer is a slot
of a pinned ring (plugins/samplesource/limesdrinput/limesdrinputthread.cpp:77-135: LMS_RecvStream(buf) -> decimate ->
SampleSinkFifo::write).  Every block must come out exactly as a separate decimateK_x(&it, buf, len) call would produce it
-- including a short last block with its own tail-drop -- whatever the coalescing factor."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flush", [1, 3, 7])
@pytest.mark.parametrize("log2,fcpos", [(6, sa.FC_CEN), (4, sa.FC_INF)])
def test_ring_blocks_equal_separate_calls(log2, fcpos, flush):
    blk = 32768                                      # LimeSDR block (complex samples)
    n_blocks = 19
    x = orc.synth_iq(n_blocks * blk, seed=31 + flush, amp=2047, tone=(0.0031, 800))
    g = sa.Decimators(log2, fcpos, 12)
    o = orc.Decim(log2, fcpos, 12)
    g.ring_create(2 * blk, 8, flush)
    got, want, in_flight = [], [], 0
    sizes = [2 * blk] * n_blocks
    sizes[11] = 2 * 20000 + 6                        # a short block in the middle: flushed alone, its tail dropped
    sizes[-1] = 2 * 1000
    pos = 0
    for nb in sizes:
        if in_flight == 8 - 1:                       # keep one slot free: retire the oldest
            got.append(g.ring_retire().copy()); in_flight -= 1
        slot = g.ring_acquire()
        slot[:nb] = x[pos: pos + nb]
        g.ring_submit(nb)
        want.append(o.process(x[pos: pos + nb]))
        pos += nb; in_flight += 1
    while in_flight:
        got.append(g.ring_retire().copy()); in_flight -= 1
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.size == b.size and np.array_equal(a, b), (i, a.size, b.size)
    with pytest.raises(sa.SdrxError):
        g.ring_retire()                              # nothing left


def test_ring_full_and_misuse_are_reported():
    g = sa.Decimators(3, sa.FC_CEN, 12)
    with pytest.raises(sa.SdrxError):
        g.ring_create(2 * 100 + 2, 4, 1)             # not a whole number of groups
    g.ring_create(2 * 4096, 2, 1)
    for _ in range(2):
        g.ring_acquire()[:] = 0
        g.ring_submit(2 * 4096)
    with pytest.raises(sa.SdrxError):
        g.ring_acquire()                             # both slots un-retired
    g.ring_retire(); g.ring_retire()
    with pytest.raises(sa.SdrxError):
        g.ring_submit(16)                            # nothing acquired
