"""Deterministic synthetic int16 I/Q for fixtures and tests: pure integer arithmetic (a 64-bit LCG),
so the same bytes come out on every numpy version / platform -- no RNG-stream stability assumed."""
import numpy as np

_A = np.uint64(6364136223846793005)
_C = np.uint64(1442695040888963407)


def lcg_u32(n: int, seed: int) -> np.ndarray:
    """n pseudo-random uint32 (top half of a 64-bit LCG state), vectorised by jump-ahead in blocks."""
    out = np.empty(n, np.uint32)
    s = np.uint64(seed * 2 + 1)
    # sequential in chunks of 1: too slow in python -> generate with a vectorised leapfrog of 4096 lanes
    lanes = 4096
    st = np.empty(lanes, np.uint64)
    with np.errstate(over="ignore"):
        for i in range(lanes):
            s = s * _A + _C
            st[i] = s
        # multiplier/increment for a jump of `lanes` steps
        a, c = np.uint64(1), np.uint64(0)
        for _ in range(lanes):
            c = c * _A + _C
            a = a * _A
        pos = 0
        while pos < n:
            m = min(lanes, n - pos)
            out[pos:pos + m] = (st[:m] >> np.uint64(32)).astype(np.uint32)
            st = st * a + c
            pos += m
    return out


def noise_iq(n_cplx: int, seed: int, amp: int) -> np.ndarray:
    """uniform int16 I/Q in [-amp, amp] (interleaved), amp <= 32767"""
    u = lcg_u32(2 * n_cplx, seed).astype(np.int64)
    return ((u % (2 * amp + 1)) - amp).astype(np.int16)


# one period of an integer "tone" at fs/16: round(1000*cos/sin(2*pi*k/16)), hard-coded so no libm is involved
_C16 = np.array([1000, 924, 707, 383, 0, -383, -707, -924, -1000, -924, -707, -383, 0, 383, 707, 924], np.int64)
_S16 = np.roll(_C16, 4)


def tone_iq(n_cplx: int, amp_milli: int, step: int = 1) -> np.ndarray:
    """complex tone at step*fs/16 with amplitude amp_milli (table scaled by amp_milli/1000, integer division)"""
    k = (np.arange(n_cplx, dtype=np.int64) * step) % 16
    x = np.empty(2 * n_cplx, np.int64)
    x[0::2] = _C16[k] * amp_milli // 1000
    x[1::2] = _S16[k] * amp_milli // 1000
    return x


def mix(n_cplx: int, seed: int, amp: int, tone_amp: int = 0, step: int = 1) -> np.ndarray:
    x = noise_iq(n_cplx, seed, amp).astype(np.int64)
    if tone_amp:
        x = x + tone_iq(n_cplx, tone_amp, step)
    return np.clip(x, -32768, 32767).astype(np.int16)


def fnv1a64(a: np.ndarray) -> int:
    """FNV-1a over the little-endian bytes of a (used for the big golden outputs)"""
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


# float half-band decimator fixtures (tests/golden/fdecim_golden.npz): inputs are integer-generated and scaled by an
# exact power of two, so every platform rebuilds the same float32 bits
FDECIM_CASES = [(0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2), (3, 0), (3, 1), (3, 2), (4, 1), (4, 2), (5, 0), (6, 0), (6, 1), (6, 2)]


def fdecim_input(kind, n_cplx, seed):
    x = mix(n_cplx, seed, 2000, 900, 1)                    # int16 I/Q, |x| < 2048
    if kind.startswith("if"):
        return x
    return (x.astype(np.float32) / np.float32(4096.0)).astype(np.float32)      # |x| < 0.5, exact


def fdecim_cuts(n_cplx):
    c = [0, 1000, 1001, n_cplx // 2 + 3, n_cplx]
    return list(zip(c[:-1], c[1:]))


# ---- inputs of tests/golden/wide24_golden.* (the reference's SDR_RX_SAMPLE_24BIT build)
W24_DEC_N = 12288 + 200
W24_DEC_CUTS = [0, 0, 6, 2 * 2048 + 2, 2 * 5001, 2 * 9000 + 128, 2 * W24_DEC_N]
W24_CH_N = 1 << 15
W24_CH_CUTS = [0, 5, 4099, 4099, 20000, 20001, W24_CH_N]
W24_CH_MODES = ([0], [1], [2], [1, 2, 0], [2, 2, 1, 0, 1], [0, 0, 0, 0, 0, 0, 0], [1, 0, 2, 1, 0, 2, 1, 0, 2, 1, 0, 2], [2, 1, 1, 0, 2, 0, 1, 2, 2, 0, 1, 1, 0])


def w24_dec_inputs():
    n = W24_DEC_N
    w = np.empty(2 * n, np.int16); w[0::2] = -32768; w[1::2] = np.where(np.arange(n) % 3 == 0, 32767, -32768)
    w[:4000] = noise_iq(2000, 14, 32767)
    return {"b12": mix(n, 11, 2047, 900, 1), "b8": mix(n, 12, 127, 60, 1), "b16": mix(n, 13, 32767, 0), "wrap": w}


def w24_chan_inputs():
    """24-bit samples: a 16-bit noise word times 256 plus a second noise byte; `full` pins every 7th int32 to -2^23"""
    hi = noise_iq(W24_CH_N, 41, 32767).astype(np.int32); lo = noise_iq(W24_CH_N, 42, 127).astype(np.int32)
    full = hi * 256 + lo
    full[::7] = -(1 << 23)
    return {"n24": mix(W24_CH_N, 43, 32767, 9000, 3).astype(np.int32) * 64, "full": full}


def noise24(n_cplx: int, seed: int) -> np.ndarray:
    """full-range 24-bit I/Q as interleaved int32"""
    return noise_iq(n_cplx, seed, 32767).astype(np.int32) * 256 + noise_iq(n_cplx, seed + 7919, 127).astype(np.int32)
