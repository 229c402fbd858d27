"""GPU parity of sdrx_decim_process_dev_batch: many device streams (one Decimators object per device thread in the
reference, plugins/samplesource/limesdrinput/limesdrinputthread.cpp:103-135), ONE launch, every stream bit-exact
against its own oracle object, state carried per stream across batched calls, ragged and empty streams included."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["1", "4"], autouse=True)
def _fast_flavour(request, monkeypatch):
    """both flavours of the FAST kernel: single-wave workgroups (long launches) and four-wave workgroups (short ones);
    the library picks by launch size, the tests pin each in turn"""
    monkeypatch.setenv("SDRX_DECIM_NW", request.param)
torch = pytest.importorskip("torch")


def run_batch(handles, xs, ranges):
    """one batched call over xs[i][a:b] (int16 element ranges); returns per-stream outputs"""
    segs = [np.ascontiguousarray(x[a:b]) for x, (a, b) in zip(xs, ranges)]
    # 16-byte aligned device buffers, one allocation per stream
    d_in = [torch.from_numpy(np.concatenate([s, np.zeros(8, s.dtype)])).cuda() for s in segs]
    d_out = [torch.zeros(max(s.size, 2) + 64, dtype=torch.int16, device="cuda") for s in segs]
    torch.cuda.synchronize()
    n_out = sa.decimate_dev_batch(handles, [t.data_ptr() for t in d_in], [s.size for s in segs], [t.data_ptr() for t in d_out])
    handles[0].sync()
    return [o[: 2 * n].cpu().numpy() for o, n in zip(d_out, n_out)]


@pytest.mark.parametrize("log2,fcpos,bits", [(6, sa.FC_CEN, 12), (4, sa.FC_INF, 12), (6, sa.FC_SUP, 16), (1, sa.FC_CEN, 8), (0, sa.FC_CEN, 12)])
def test_batch_matches_per_stream_oracle(log2, fcpos, bits):
    n_str = 7
    amp = {8: 127, 12: 2047, 16: 32767}[bits]
    lens = [70_000, 32_768, 1, 0, 123_457, 4096, 65_536 + 12]
    xs = [orc.synth_iq(max(n, 1), seed=40 + i, amp=amp, tone=(0.001 * (i + 1), 0.4 * amp))[: 2 * n] for i, n in enumerate(lens)]
    hs = [sa.Decimators(log2, fcpos, bits) for _ in range(n_str)]
    os_ = [orc.Decim(log2, fcpos, bits) for _ in range(n_str)]
    for frac in ((0.0, 0.31), (0.31, 0.31), (0.31, 1.0)):             # three batched calls, the middle one empty everywhere
        ranges = [(2 * int(frac[0] * n), 2 * int(frac[1] * n)) for n in lens]
        got = run_batch(hs, xs, ranges)
        for i in range(n_str):
            want = os_[i].process(xs[i][ranges[i][0]: ranges[i][1]])
            assert got[i].size == want.size and np.array_equal(got[i], want), (i, frac, got[i].size, want.size)


def test_batch_of_64_lime_blocks_and_more_than_one_launch():
    """70 streams x 32 768-sample blocks (the LimeSDR thread's block): 64 go in the first launch, 6 in a second one"""
    n_str, blk = 70, 32768
    xs = [orc.synth_iq(3 * blk, seed=900 + i, amp=2047, tone=(0.0007 * (1 + i % 5), 700)) for i in range(n_str)]
    hs = [sa.Decimators(6, sa.FC_CEN, 12) for _ in range(n_str)]
    want = [orc.Decim(6, sa.FC_CEN, 12).process(x) for x in xs]
    parts = [[] for _ in range(n_str)]
    for b in range(3):
        got = run_batch(hs, xs, [(2 * b * blk, 2 * (b + 1) * blk)] * n_str)
        for i in range(n_str):
            parts[i].append(got[i])
    for i in range(n_str):
        assert np.array_equal(np.concatenate(parts[i]), want[i]), i


def test_batch_u8_flavour_and_argument_checks():
    from tests import synth
    n = 50_000
    xs = [(synth.lcg_u32(2 * n, 77 + i) & 0xff).astype(np.uint8) for i in range(3)]
    hs = [sa.DecimatorsU(5, sa.FC_CEN, 127) for _ in range(3)]
    got = run_batch(hs, xs, [(0, 2 * n)] * 3)
    for i in range(3):
        assert np.array_equal(got[i], orc.DecimU(5, sa.FC_CEN, 127).process(xs[i])), i
    other = sa.Decimators(5, sa.FC_CEN, 12)
    with pytest.raises(sa.SdrxError):
        run_batch([hs[0], other], xs[:2], [(0, 64)] * 2)              # mixed configurations
    with pytest.raises(sa.SdrxError):
        run_batch([hs[0], hs[0]], xs[:2], [(0, 64)] * 2)              # the same handle twice
