"""The end-to-end example (examples/filesource_to_nfm.py: .sdriq -> FIFO -> DC correction -> channelizer bank -> NFM front) runs on
the GPU and recovers the tone each synthetic FM carrier was modulated with."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_filesource_replay_recovers_the_modulating_tones(tmp_path):
    spec = importlib.util.spec_from_file_location("filesource_to_nfm", os.path.join(ROOT, "examples", "filesource_to_nfm.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    audio = mod.main(str(tmp_path))
    for c, parts in enumerate(audio):
        y = np.concatenate(parts)[2000:]
        s = np.abs(np.fft.rfft(y - y.mean()))
        f_peak = np.argmax(s) * 48000.0 / (2 * (s.size - 1))
        assert abs(f_peak - (700 + 300 * c)) < 20.0, (c, f_peak)
    # second leg: the qint16 audio out of the NFM tail (squelch open on every carrier) carries the same tone
    for c, parts in enumerate(mod.main.pcm):
        z = np.concatenate(parts).astype(np.float64)[4000:]
        assert np.abs(z).max() > 100, c                      # the squelch opened
        s = np.abs(np.fft.rfft(z - z.mean()))
        f_peak = np.argmax(s) * 48000.0 / (2 * (s.size - 1))
        assert abs(f_peak - (700 + 300 * c)) < 20.0, (c, f_peak)
