"""GPU: random sequences of bank operations (feeds of any length, partial reads, skips, reconfigure, add / remove channel,
reset) against a model made of oracle chains and Python queues -- every sample of every channel, after every operation."""
import os

import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc
from tests import synth

pytestmark = pytest.mark.gpu
FS = 2_400_000
RATES = [FS, FS // 2, 300_000, 48_000, 12_500, 2_000, 300]


class Model:
    def __init__(self, rate, fc):
        self.configure(rate, fc)
        self.q = np.zeros(0, np.int16)
        self.dead = False

    def configure(self, rate, fc):
        self.modes, self.out_rate, self.ofs = orc.chan_plan(FS, rate, fc)
        self.chain = orc.Chain(self.modes) if len(self.modes) else None
        self.dead = False

    def feed(self, seg):
        if self.dead:
            return
        y = self.chain.feed(seg) if self.chain is not None else seg
        self.q = np.concatenate([self.q, y])


def _rand_cfg(rng):
    rate = int(rng.choice(RATES))
    return rate, int(rng.integers(-FS // 2 + rate // 2, FS // 2 - rate // 2 + 1))


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDRX_FUZZ_SEEDS", "6"))))      # a soak run sets SDRX_FUZZ_SEEDS=200
def test_random_operation_sequences(seed):
    rng = np.random.default_rng(1000 + seed)
    x = synth.noise_iq(600_000, 70 + seed, 32767)
    x[::9] = -32768
    pos = 0
    cfgs = [_rand_cfg(rng) for _ in range(int(rng.integers(1, 6)))]
    bank = sa.ChannelizerBank(FS, [c[0] for c in cfgs], [c[1] for c in cfgs])
    model = [Model(*c) for c in cfgs]

    def check_all(tag):
        for c, m in enumerate(model):
            got = bank.available(c)
            assert got == m.q.size // 2, (seed, tag, c, got, m.q.size // 2)

    for op_i in range(60):
        op = rng.choice(["feed", "feed", "feed", "read", "read", "skip", "reconf", "add", "remove", "reset"], p=[.2, .15, .1, .15, .1, .05, .1, .06, .05, .04])
        live = [c for c, m in enumerate(model) if not m.dead]
        if op == "feed":
            n = int(rng.choice([0, 1, 7, 255, 4095, 4096, 4097, 9000, 20000, 50001]))
            if pos + n > 600_000:
                pos = 0
            seg = x[2 * pos: 2 * (pos + n)]; pos += n
            bank.feed(seg)
            for m in model:
                m.feed(seg)
        elif op == "read" and live:
            c = int(rng.choice(live)); m = model[c]
            have = m.q.size // 2
            cap = have if rng.random() < 0.5 else int(rng.integers(0, have + 1))
            got = bank.read(c, cap)
            assert np.array_equal(got, m.q[: 2 * cap]), (seed, op_i, "read", c, cap)
            m.q = m.q[2 * cap:]
        elif op == "skip" and live:
            c = int(rng.choice(live)); m = model[c]
            have = m.q.size // 2
            n = int(rng.integers(0, have + 1))
            bank.skip(c, n)
            m.q = m.q[2 * n:]
        elif op == "reconf" and model:
            c = int(rng.integers(0, len(model)))
            rate, fc = _rand_cfg(rng)
            bank.reconfigure(c, rate, fc)
            model[c].configure(rate, fc)              # fresh chain, zero history; what was queued stays queued
            mo, r, o = bank.info(c)
            assert list(mo) == list(model[c].modes) and (r, o) == (model[c].out_rate, model[c].ofs)
        elif op == "add" and len(model) < 10:
            rate, fc = _rand_cfg(rng)
            c = bank.add_channel(rate, fc)
            assert c == len(model)
            model.append(Model(rate, fc))
        elif op == "remove" and live:
            c = int(rng.choice(live))
            bank.remove_channel(c)
            model[c].dead = True; model[c].q = np.zeros(0, np.int16)
        elif op == "reset":
            bank.reset()
            for m in model:
                if not m.dead:
                    m.chain = orc.Chain(m.modes) if len(m.modes) else None
                m.q = np.zeros(0, np.int16)
        check_all((op_i, op))
    for c, m in enumerate(model):                       # drain
        if not m.dead:
            assert np.array_equal(bank.read(c), m.q), (seed, "drain", c)
    bank.close()
