#!/usr/bin/env python3
"""Generate tests/golden/wide24_golden.{npz,json} FROM THE REFERENCE'S 24-BIT BUILD (build container only).

`make -C oracle ref24` compiles the reference's own headers with -DSDR_RX_SAMPLE_24BIT into oracle/_ref/libsdrref24.so
(wrapper: oracle/ref_shim24.cpp); this script runs Decimators<qint32,qint16,24,{8,12,16}> and DownChannelizer-style
IntHalfbandFilterEO<qint64,qint64,48> chains of that build on deterministic inputs (tests/synth.py) and stores the
expected outputs (verbatim for a few cases, FNV-1a hashes for all).  Data only; no reference source travels.

    python tests/golden/make_golden24.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsdrref24.so"))
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
ref.ref24_decim_new.restype = vp; ref.ref24_decim_new.argtypes = [C.c_int]
ref.ref24_decim_free.argtypes = [vp]
ref.ref24_decim_process.restype = C.c_int; ref.ref24_decim_process.argtypes = [vp, C.c_int, C.c_int, vp, i32, vp]
ref.ref24_chain_new.restype = vp; ref.ref24_chain_new.argtypes = [C.c_int, vp]
ref.ref24_chain_free.argtypes = [vp]
ref.ref24_chain_feed.restype = i64; ref.ref24_chain_feed.argtypes = [vp, vp, i64, vp]
assert ref.ref24_sample_bytes() == 8

DEC_N, DEC_CUTS, CH_N, CH_CUTS, CH_MODES = synth.W24_DEC_N, synth.W24_DEC_CUTS, synth.W24_CH_N, synth.W24_CH_CUTS, synth.W24_CH_MODES
dec_inputs, chan_inputs = synth.w24_dec_inputs, synth.w24_chan_inputs


def main():
    hashes, keep = {}, {}
    for name, x in dec_inputs().items():
        for bits in (8, 12, 16):
            for log2 in range(7):
                for fc in range(3):
                    h = ref.ref24_decim_new(bits); outs = []
                    for a, b in zip(DEC_CUTS[:-1], DEC_CUTS[1:]):
                        seg = np.ascontiguousarray(x[a:b]); o = np.zeros(seg.size + 16, np.int32)
                        n = ref.ref24_decim_process(h, log2, fc, seg.ctypes.data, seg.size, o.ctypes.data)
                        outs.append(o[: 2 * n].copy())
                    ref.ref24_decim_free(h)
                    y = np.concatenate(outs)
                    key = f"dec_{name}_bits{bits}_log{log2}_fc{fc}"
                    hashes[key] = {"n": int(y.size // 2), "fnv1a64": f"{synth.fnv1a64(y):016x}"}
                    if log2 >= 5 and (name == "wrap" or bits == 12):
                        keep[key] = y
    for name, x in chan_inputs().items():
        for modes in CH_MODES:
            m = np.ascontiguousarray(modes, dtype=np.uint8)
            cuts = CH_CUTS if len(modes) > 3 else CH_CUTS[:4]
            h = ref.ref24_chain_new(m.size, m.ctypes.data); outs = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                seg = np.ascontiguousarray(x[2 * a: 2 * b]); o = np.zeros(seg.size + 16, np.int32)
                n = ref.ref24_chain_feed(h, seg.ctypes.data, b - a, o.ctypes.data)
                outs.append(o[: 2 * n].copy())
            ref.ref24_chain_free(h)
            y = np.concatenate(outs)
            key = f"chain_{name}_{''.join(map(str, modes))}"
            hashes[key] = {"n": int(y.size // 2), "fnv1a64": f"{synth.fnv1a64(y):016x}"}
            if len(modes) >= 5:
                keep[key] = y
    np.savez_compressed(os.path.join(HERE, "wide24_golden.npz"), **keep)
    json.dump({"recipe": "inputs: tests/synth.py w24_dec_inputs()/w24_chan_inputs(); cuts W24_DEC_CUTS (int16 units) / W24_CH_CUTS (samples)", "hashes": hashes},
              open(os.path.join(HERE, "wide24_golden.json"), "w"), indent=0, sort_keys=True)
    print(len(hashes), "cases,", len(keep), "verbatim")


if __name__ == "__main__":
    main()
