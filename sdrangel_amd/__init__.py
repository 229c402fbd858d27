"""sdrangel_amd -- ctypes face of libsdrx.so (the MI355X engine for SDRangel's sdrbase/dsp RX path).

The product is the C-ABI shared library (include/sdrx.h); this module only loads it and gives
tests / bench.py numpy-friendly wrappers whose names follow the reference classes
(`Decimators`, `DownChannelizer` bank, `SampleSinkFifo`).  There is no CPU fallback: if the
library is missing, or no HIP device is present when a GPU object is created, it raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdrx.so")

FC_INF, FC_SUP, FC_CEN = 0, 1, 2
MODE_CENTER, MODE_LOWER, MODE_UPPER = 0, 1, 2

_lib = None


class SdrxError(RuntimeError):
    pass


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels carry a private libamdhip64.so with the
    same SONAME (libamdhip64.so.7) as /opt/rocm's.  If libsdrx.so pulled in ROCm's copy first and
    torch its own later, the process would hold two runtimes and the second would see no GPU.
    Loading torch's copy first (by path, without importing torch) makes both resolve to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def _del(self):
    """Finaliser shared by the handle classes: close(), quietly -- at interpreter shutdown the module's globals may already be gone."""
    try:
        self.close()
    except Exception:
        pass


def lib() -> C.CDLL:
    """Load libsdrx.so (built in-tree by `make -C sdrangel_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdrxError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    pp = C.POINTER(C.c_void_p)
    sig = {
        "sdrx_version": (C.c_char_p, []),
        "sdrx_last_error": (C.c_char_p, []),
        "sdrx_device_count": (C.c_int, []),
        "sdrx_decim_create": (C.c_int, [pp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sdrx_decim_create_u8": (C.c_int, [pp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sdrx_decim_process_u8": (C.c_int, [vp, vp, i32, vp, C.POINTER(i32)]),
        "sdrx_decim_process_dev_u8": (C.c_int, [vp, vp, i64, vp, C.POINTER(i64)]),
        "sdrx_decim_destroy": (C.c_int, [vp]),
        "sdrx_decim_reset": (C.c_int, [vp]),
        "sdrx_decim_process": (C.c_int, [vp, vp, i32, vp, C.POINTER(i32)]),
        "sdrx_decim_process_dev": (C.c_int, [vp, vp, i64, vp, C.POINTER(i64)]),
        "sdrx_decim_process_dev_batch": (C.c_int, [vp, i32, vp, vp, vp, vp]),
        "sdrx_decim_sync": (C.c_int, [vp]),
        "sdrx_decim_ring_create": (C.c_int, [vp, i32, i32, i32]),
        "sdrx_decim_ring_destroy": (C.c_int, [vp]),
        "sdrx_decim_ring_acquire": (vp, [vp]),
        "sdrx_decim_ring_submit": (C.c_int, [vp, i32]),
        "sdrx_decim_ring_retire": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i32)]),
        "sdrx_decim_set_stream": (C.c_int, [vp, vp]),
        "sdrx_decim_group_int16": (C.c_int, [C.c_int, C.c_int]),
        "sdrx_decim_state_bytes": (i64, [vp]),
        "sdrx_decim_get_state": (C.c_int, [vp, vp]),
        "sdrx_decim_set_state": (C.c_int, [vp, vp]),
        "sdrx_decim_set_timing": (C.c_int, [vp, C.c_int]),
        "sdrx_decim_get_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.c_int]),
        "sdrx_decim_last_launch": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "sdrx_chan_bank_create": (C.c_int, [pp, C.c_int, i32, i32, vp, vp]),
        "sdrx_chan_bank_destroy": (C.c_int, [vp]),
        "sdrx_chan_bank_info": (C.c_int, [vp, i32, C.POINTER(i32), vp, C.POINTER(i32), C.POINTER(i32)]),
        "sdrx_chan_plan": (C.c_int, [i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]),
        "sdrx_chan_bank_reconfigure": (C.c_int, [vp, i32, i32, i32]),
        "sdrx_chan_bank_reset": (C.c_int, [vp]),
        "sdrx_chan_bank_add_channel": (C.c_int, [vp, i32, i32, C.POINTER(i32)]),
        "sdrx_chan_bank_remove_channel": (C.c_int, [vp, i32]),
        "sdrx_chan_bank_group_count": (i32, [vp]),
        "sdrx_chan_bank_feed": (C.c_int, [vp, vp, i64]),
        "sdrx_chan_bank_feed_dev": (C.c_int, [vp, vp, i64]),
        "sdrx_chan_bank_available": (i64, [vp, i32]),
        "sdrx_chan_bank_read": (i64, [vp, i32, vp, i64]),
        "sdrx_chan_bank_skip": (i64, [vp, i32, i64]),
        "sdrx_chan_bank_last_dev": (C.c_int, [vp, i32, pp, C.POINTER(i64)]),
        "sdrx_chan_bank_sync": (C.c_int, [vp]),
        "sdrx_chan_bank_set_stream": (C.c_int, [vp, vp]),
        "sdrx_chan_bank_get_stream": (C.c_int, [vp, C.POINTER(vp)]),
        "sdrx_chan_bank_set_timing": (C.c_int, [vp, C.c_int]),
        "sdrx_chan_bank_get_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.c_int]),
        "sdrx_chan_bank_last_launch": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "sdrx_backend_create": (C.c_int, [pp, C.c_int, i32, vp]),
        "sdrx_backend_destroy": (C.c_int, [vp]),
        "sdrx_backend_feed": (C.c_int, [vp, vp, vp]),
        "sdrx_backend_feed_dev": (C.c_int, [vp, vp, vp]),
        "sdrx_backend_feed_bank": (C.c_int, [vp, vp]),
        "sdrx_measure_hbm_read": (C.c_int, [C.c_int, C.c_uint64, C.c_int32, C.POINTER(C.c_double)]),
        "sdrx_dccorr_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "sdrx_dccorr_destroy": (C.c_int, [vp]),
        "sdrx_dccorr_reset": (C.c_int, [vp]),
        "sdrx_dccorr_process": (C.c_int, [vp, vp, C.c_int64]),
        "sdrx_dccorr_process_dev": (C.c_int, [vp, vp, vp, C.c_int64]),
        "sdrx_dccorr_sync": (C.c_int, [vp]),
        "sdrx_dccorr_set_stream": (C.c_int, [vp, vp]),
        "sdrx_iqimb_create": (C.c_int, [C.POINTER(vp), C.c_int, i32]),
        "sdrx_iqimb_destroy": (C.c_int, [vp]),
        "sdrx_iqimb_reset": (C.c_int, [vp]),
        "sdrx_iqimb_process": (C.c_int, [vp, vp, vp]),
        "sdrx_iqimb_process_dev": (C.c_int, [vp, vp, vp, vp]),
        "sdrx_iqimb_sync": (C.c_int, [vp]),
        "sdrx_iqimb_set_stream": (C.c_int, [vp, vp]),
        "sdrx_fdecim_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sdrx_fdecim_destroy": (C.c_int, [vp]),
        "sdrx_fdecim_reset": (C.c_int, [vp]),
        "sdrx_fdecim_process": (C.c_int, [vp, vp, C.c_int32, vp, C.POINTER(C.c_int32)]),
        "sdrx_fdecim_process_dev": (C.c_int, [vp, vp, C.c_int64, vp, C.POINTER(C.c_int64)]),
        "sdrx_fdecim_sync": (C.c_int, [vp]),
        "sdrx_fdecim_set_stream": (C.c_int, [vp, vp]),
        "sdrx_fdecim_group": (C.c_int32, [C.c_int, C.c_int]),
        "sdrx_fdecim_set_timing": (C.c_int, [vp, C.c_int]),
        "sdrx_fdecim_get_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
        "sdrx_fdecim_last_launch": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "sdrx_backend_read": (i64, [vp, i32, vp, i64]),
        "sdrx_backend_get_design": (C.c_int, [vp, i32, C.POINTER(i32), vp, i32, vp, C.POINTER(i32)]),
        "sdrx_backend_sync": (C.c_int, [vp]),
        "sdrx_audiotail_create": (C.c_int, [pp, C.c_int, i32, vp]),
        "sdrx_audiotail_destroy": (C.c_int, [vp]),
        "sdrx_audiotail_reset": (C.c_int, [vp]),
        "sdrx_audiotail_feed": (C.c_int, [vp, vp, vp, vp]),
        "sdrx_audiotail_feed_dev": (C.c_int, [vp, vp, vp, vp]),
        "sdrx_audiotail_sync": (C.c_int, [vp]),
        "sdrx_iir_create": (C.c_int, [pp, C.c_int, i32, vp]),
        "sdrx_iir_destroy": (C.c_int, [vp]),
        "sdrx_iir_reset": (C.c_int, [vp]),
        "sdrx_iir_feed": (C.c_int, [vp, vp, vp, vp]),
        "sdrx_decim24_create": (C.c_int, [pp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sdrx_decim24_destroy": (C.c_int, [vp]),
        "sdrx_decim24_reset": (C.c_int, [vp]),
        "sdrx_decim24_process": (C.c_int, [vp, vp, i32, vp, C.POINTER(i32)]),
        "sdrx_decim_stages_create": (C.c_int, [pp, C.c_int]),
        "sdrx_decim_stages_destroy": (C.c_int, [vp]),
        "sdrx_decim_save_stages": (C.c_int, [vp, vp]),
        "sdrx_decim_load_stages": (C.c_int, [vp, vp]),
        "sdrx_fanout_create": (C.c_int, [pp, C.c_int, i32, vp, i64]),
        "sdrx_fanout_destroy": (C.c_int, [vp]),
        "sdrx_fanout_send": (C.c_int, [vp, vp, i64, vp]),
        "sdrx_fanout_buffer": (vp, [vp, i32]),
        "sdrx_fanout_wait": (C.c_int, [vp, i32]),
        "sdrx_fanout_stream_wait": (C.c_int, [vp, i32, vp]),
        "sdrx_fdecim_state_bytes": (i64, [vp]),
        "sdrx_fdecim_get_state": (C.c_int, [vp, vp]),
        "sdrx_fdecim_set_state": (C.c_int, [vp, vp]),
        "sdrx_chan_bank_state_bytes": (i64, [vp]),
        "sdrx_chan_bank_get_state": (C.c_int, [vp, vp]),
        "sdrx_chan_bank_set_state": (C.c_int, [vp, vp]),
        "sdrx_fdecim_stages_create": (C.c_int, [pp, C.c_int]),
        "sdrx_fdecim_stages_destroy": (C.c_int, [vp]),
        "sdrx_fdecim_save_stages": (C.c_int, [vp, vp]),
        "sdrx_fdecim_load_stages": (C.c_int, [vp, vp]),
        "sdrx_decim24_process_dev": (C.c_int, [vp, vp, i64, vp, C.POINTER(i64)]),
        "sdrx_decim24_sync": (C.c_int, [vp]),
        "sdrx_chan24_bank_feed_dev": (C.c_int, [vp, vp, i64]),
        "sdrx_chan24_bank_out_dev": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(i64)]),
        "sdrx_chan24_bank_sync": (C.c_int, [vp]),
        "sdrx_chan24_bank_create": (C.c_int, [pp, C.c_int, i32, i32, vp, vp]),
        "sdrx_chan24_bank_destroy": (C.c_int, [vp]),
        "sdrx_chan24_bank_reset": (C.c_int, [vp]),
        "sdrx_chan24_bank_info": (C.c_int, [vp, i32, C.POINTER(i32), vp, C.POINTER(i32), C.POINTER(i32)]),
        "sdrx_chan24_bank_feed": (C.c_int, [vp, vp, i64]),
        "sdrx_chan24_bank_read": (i64, [vp, i32, vp, i64]),
        "sdrx_firbank_create": (C.c_int, [pp, C.c_int, i32, vp]),
        "sdrx_firbank_destroy": (C.c_int, [vp]),
        "sdrx_firbank_feed": (C.c_int, [vp, vp, vp, vp]),
        "sdrx_firbank_get_taps": (C.c_int, [vp, i32, vp, i32]),
        "sdrx_sdriq_parse_header": (C.c_int, [vp, C.c_uint64, vp]),
        "sdrx_sdriq_write_header": (C.c_int, [vp, vp]),
        "sdrx_fifo_create": (C.c_int, [pp, u32]),
        "sdrx_fifo_destroy": (C.c_int, [vp]),
        "sdrx_fifo_set_size": (C.c_int, [vp, u32]),
        "sdrx_fifo_size": (u32, [vp]),
        "sdrx_fifo_fill": (u32, [vp]),
        "sdrx_fifo_write_bytes": (u32, [vp, vp, u32]),
        "sdrx_fifo_write": (u32, [vp, vp, u32]),
        "sdrx_fifo_read": (u32, [vp, vp, u32]),
        "sdrx_fifo_read_begin": (u32, [vp, u32, pp, C.POINTER(u32), pp, C.POINTER(u32)]),
        "sdrx_fifo_read_commit": (u32, [vp, u32]),
        "sdrx_fifo_dropped": (C.c_uint64, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here == the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


#: every symbol include/sdrx.h declares (checked against the built library by tests/test_abi.py)
def exported_symbols(header_path: str | None = None) -> list[str]:
    import re
    header_path = header_path or os.path.join(os.path.dirname(_HERE), "include", "sdrx.h")
    txt = open(header_path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrx_[a-z0-9_]+)\s*\(", txt)) - {"sdrx_fifo_data_ready_cb"})


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise SdrxError(f"{what}: rc={rc}: {lib().sdrx_last_error().decode()}")


def _i16(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype != np.int16:
        raise TypeError("expected int16 I/Q")
    return a


class Decimators:
    """Decimators<qint32, qint16, 16, input_bits> used with one (log2, fcpos)
    (sdrbase/dsp/decimators.h:279-341).  `decimate(buf)` == decimateK_{inf,sup,cen}(&it, buf, len)."""

    def __init__(self, log2_decim: int, fcpos: int = FC_CEN, input_bits: int = 12, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().sdrx_decim_create(C.byref(self._h), device, log2_decim, fcpos, input_bits), "sdrx_decim_create")
        self.log2, self.fcpos, self.input_bits = log2_decim, fcpos, input_bits

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_decim_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_decim_reset(self._h), "sdrx_decim_reset")

    def decimate(self, buf) -> np.ndarray:
        """buf: int16 interleaved I,Q (len = reference `len`).  Returns int16 array of 2*n_out."""
        buf = _i16(buf)
        out = np.empty(max(2 * ((buf.size // 2) >> self.log2), 2), np.int16)
        n = C.c_int32()
        _check(lib().sdrx_decim_process(self._h, buf.ctypes.data, buf.size, out.ctypes.data, C.byref(n)), "sdrx_decim_process")
        return out[: 2 * n.value]

    def decimate_dev(self, d_in_ptr: int, n_int16: int, d_out_ptr: int) -> int:
        """device pointers; asynchronous on the handle's stream; returns #complex outputs"""
        n = C.c_int64()
        _check(lib().sdrx_decim_process_dev(self._h, d_in_ptr, n_int16, d_out_ptr, C.byref(n)), "sdrx_decim_process_dev")
        return n.value

    def sync(self):
        _check(lib().sdrx_decim_sync(self._h), "sdrx_decim_sync")

    def save_stages(self, stages: "DecimStages"):
        """stages 1..log2 of `stages` := what this variant's filters hold now"""
        _check(lib().sdrx_decim_save_stages(self._h, stages._h), "sdrx_decim_save_stages")

    def load_stages(self, stages: "DecimStages"):
        """continue from the shared filter set of one reference Decimators object (decimators.h:326-333)"""
        _check(lib().sdrx_decim_load_stages(self._h, stages._h), "sdrx_decim_load_stages")

    # ---- pinned double-buffered host path: the receive buffer IS a slot of the handle's pinned ring
    def ring_create(self, slot_elems: int, n_slots: int, flush_slots: int = 1):
        _check(lib().sdrx_decim_ring_create(self._h, slot_elems, n_slots, flush_slots), "sdrx_decim_ring_create")
        self._slot_elems = slot_elems

    def ring_acquire(self) -> np.ndarray:
        """numpy view of the next free pinned input slot (int16, or uint8 for DecimatorsU)"""
        p = lib().sdrx_decim_ring_acquire(self._h)
        if not p:
            raise SdrxError(f"sdrx_decim_ring_acquire: {lib().sdrx_last_error().decode()}")
        ct = C.c_uint8 if self.input_bits == 8 and isinstance(self, DecimatorsU) else C.c_int16
        return np.ctypeslib.as_array((ct * self._slot_elems).from_address(p))

    def ring_submit(self, n_elems: int):
        _check(lib().sdrx_decim_ring_submit(self._h, n_elems), "sdrx_decim_ring_submit")

    def ring_retire(self) -> np.ndarray:
        """outputs of the oldest submitted block (a view of the pinned output slot: copy it before that slot is reused)"""
        out, n = C.c_void_p(), C.c_int32()
        _check(lib().sdrx_decim_ring_retire(self._h, C.byref(out), C.byref(n)), "sdrx_decim_ring_retire")
        if n.value == 0:
            return np.empty(0, np.int16)
        return np.ctypeslib.as_array((C.c_int16 * (2 * n.value)).from_address(out.value))

    def set_stream(self, hip_stream: int | None):
        _check(lib().sdrx_decim_set_stream(self._h, hip_stream), "sdrx_decim_set_stream")

    def set_timing(self, on: bool):
        _check(lib().sdrx_decim_set_timing(self._h, int(on)), "sdrx_decim_set_timing")

    def get_timing(self, reset: bool = True):
        """(total kernel ms, launches) measured with HIP events on the launch stream"""
        ms, n = C.c_double(), C.c_int64()
        _check(lib().sdrx_decim_get_timing(self._h, C.byref(ms), C.byref(n), int(reset)), "sdrx_decim_get_timing")
        return ms.value, n.value

    def get_state(self) -> bytes:
        nb = lib().sdrx_decim_state_bytes(self._h)
        buf = C.create_string_buffer(nb)
        _check(lib().sdrx_decim_get_state(self._h, buf), "sdrx_decim_get_state")
        return buf.raw

    def set_state(self, state: bytes):
        if len(state) != lib().sdrx_decim_state_bytes(self._h):
            raise ValueError("state size")
        _check(lib().sdrx_decim_set_state(self._h, state), "sdrx_decim_set_state")

    def last_launch(self) -> dict:
        name = C.create_string_buffer(128)
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        _check(lib().sdrx_decim_last_launch(self._h, name, 128, C.byref(g), C.byref(b), C.byref(l)), "last_launch")
        return {"kernel": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}


def decimate_dev_batch(handles, d_in_ptrs, n_elems, d_out_ptrs) -> list:
    """sdrx_decim_process_dev_batch: many device streams (one Decimators / DecimatorsU object each, same configuration),
    one launch.  Device pointers; asynchronous on handles[0]'s stream; returns #complex outputs per stream."""
    n = len(handles)
    if not (n == len(d_in_ptrs) == len(n_elems) == len(d_out_ptrs)):
        raise ValueError("one pointer / count per handle")
    hs = (C.c_void_p * n)(*[h._h.value for h in handles])
    ins = (C.c_void_p * n)(*d_in_ptrs)
    outs = (C.c_void_p * n)(*d_out_ptrs)
    ns = (C.c_int64 * n)(*n_elems)
    no = (C.c_int64 * n)()
    _check(lib().sdrx_decim_process_dev_batch(hs, n, ins, ns, outs, no), "sdrx_decim_process_dev_batch")
    return list(no)


class DecimStages:
    """The six IntHalfbandFilterEO states that all decimateK_x of ONE reference Decimators object share."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().sdrx_decim_stages_create(C.byref(self._h), device), "sdrx_decim_stages_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_decim_stages_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del


class DecimatorsObject:
    """One reference Decimators / DecimatorsU object: any decimateK_x per call, all on the same six stage states
    (what include/sdrx/dsp.hpp's sdrx::Decimators does in C++)."""

    def __init__(self, input_bits: int = 12, device: int = 0, u8_shift=None):
        self.bits, self.device, self.u8_shift = input_bits, device, u8_shift
        self._variants, self._stages, self._last = {}, DecimStages(device), None

    def decimate(self, log2: int, fcpos: int, buf) -> np.ndarray:
        d = self._variants.get((log2, fcpos))
        if d is None:
            d = DecimatorsU(log2, fcpos, self.u8_shift, self.device) if self.u8_shift is not None else Decimators(log2, fcpos, self.bits, self.device)
            self._variants[(log2, fcpos)] = d
        if log2 > 0 and d is not self._last:
            if self._last is not None:
                self._last.save_stages(self._stages)
            d.load_stages(self._stages)
            self._last = d
        return d.decimate(buf)

    def close(self):
        for d in self._variants.values():
            d.close()
        self._variants = {}
        self._stages.close()


class DecimatorsU(Decimators):
    """DecimatorsU<qint32, quint8, 16, 8, shift> (sdrbase/dsp/decimatorsu.h): unsigned 8-bit I/Q (RTL-SDR)."""

    def __init__(self, log2_decim: int, fcpos: int = FC_CEN, shift: int = 127, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().sdrx_decim_create_u8(C.byref(self._h), device, log2_decim, fcpos, shift), "sdrx_decim_create_u8")
        self.log2, self.fcpos, self.input_bits = log2_decim, fcpos, 8

    def decimate(self, buf) -> np.ndarray:
        buf = np.ascontiguousarray(buf)
        if buf.dtype != np.uint8:
            raise TypeError("expected uint8 I/Q")
        out = np.empty(max(2 * ((buf.size // 2) >> self.log2), 2), np.int16)
        n = C.c_int32()
        _check(lib().sdrx_decim_process_u8(self._h, buf.ctypes.data, buf.size, out.ctypes.data, C.byref(n)), "sdrx_decim_process_u8")
        return out[: 2 * n.value]


class Fanout:
    """One staged source stream copied to several GPUs point-to-point (sdrx_fanout_*; xGMI peer copies on a multi-GPU node)."""

    def __init__(self, src_device: int, dst_devices, max_bytes: int):
        d = np.ascontiguousarray(dst_devices, dtype=np.int32)
        self.n = d.size
        self._h = C.c_void_p()
        _check(lib().sdrx_fanout_create(C.byref(self._h), src_device, self.n, d.ctypes.data, max_bytes), "sdrx_fanout_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_fanout_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def send(self, d_src: int, n_bytes: int, producer_stream: int | None = None):
        _check(lib().sdrx_fanout_send(self._h, d_src, n_bytes, producer_stream), "sdrx_fanout_send")

    def buffer(self, i: int) -> int:
        return lib().sdrx_fanout_buffer(self._h, i) or 0

    def wait(self, i: int):
        _check(lib().sdrx_fanout_wait(self._h, i), "sdrx_fanout_wait")


class FloatDecimStages:
    """The six IntHalfbandFilterEOF states that all decimateK_x of ONE DecimatorsFI / FF / IF object share."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().sdrx_fdecim_stages_create(C.byref(self._h), device), "sdrx_fdecim_stages_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_fdecim_stages_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del


class FloatDecimatorsObject:
    """One reference DecimatorsFI ("fi") / FF ("ff") / IF ("if") object: any decimateK_x per call on the same six filters
    (what include/sdrx/dsp.hpp's sdrx::DecimatorsFI etc. do in C++)."""

    def __init__(self, kind: str, input_bits: int = 16, device: int = 0):
        self.kind, self.bits, self.device = kind, input_bits, device
        self._variants, self._stages, self._last = {}, FloatDecimStages(device), None

    def decimate(self, log2: int, fcpos: int, buf) -> np.ndarray:
        d = self._variants.get((log2, fcpos))
        if d is None:
            d = FloatDecimators(self.kind, log2, fcpos, self.bits, self.device)
            self._variants[(log2, fcpos)] = d
        if d is not self._last:
            if self._last is not None:
                _check(lib().sdrx_fdecim_save_stages(self._last._h, self._stages._h), "sdrx_fdecim_save_stages")
            _check(lib().sdrx_fdecim_load_stages(d._h, self._stages._h), "sdrx_fdecim_load_stages")
            self._last = d
        return d.decimate(buf)

    def close(self):
        for d in self._variants.values():
            d.close()
        self._variants = {}
        self._stages.close()


class FloatDecimators:
    """The float half-band decimators over IntHalfbandFilterEOF<64>, one (log2, fcpos) per object:
    kind "fi" = DecimatorsFI (float I/Q -> int16 Sample; AirspyHF), "ff" = DecimatorsFF (float -> float),
    "if" = DecimatorsIF<qint16, input_bits> (int16 -> float).  `decimate(buf)` == decimateK_{inf,sup,cen}(&it, buf, nbIAndQ)."""

    KINDS = {"fi": (0, 0), "ff": (0, 1), "if": (1, 1)}

    def __init__(self, kind: str, log2_decim: int, fcpos: int = FC_CEN, input_bits: int = 16, device: int = 0):
        self.in_kind, self.out_kind = self.KINDS[kind]
        self._h = C.c_void_p()
        _check(lib().sdrx_fdecim_create(C.byref(self._h), device, log2_decim, fcpos, self.in_kind, self.out_kind, input_bits), "sdrx_fdecim_create")
        self.kind, self.log2, self.fcpos = kind, log2_decim, fcpos

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_fdecim_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_fdecim_reset(self._h), "sdrx_fdecim_reset")

    def decimate(self, buf) -> np.ndarray:
        buf = np.ascontiguousarray(buf, dtype=np.float32 if self.in_kind == 0 else np.int16)
        out = np.empty(buf.size + 8, np.int16 if self.out_kind == 0 else np.float32)
        n = C.c_int32()
        _check(lib().sdrx_fdecim_process(self._h, buf.ctypes.data, buf.size, out.ctypes.data, C.byref(n)), "sdrx_fdecim_process")
        return out[: 2 * n.value]

    def decimate_dev(self, d_in_ptr: int, n_elems: int, d_out_ptr: int) -> int:
        n = C.c_int64()
        _check(lib().sdrx_fdecim_process_dev(self._h, d_in_ptr, n_elems, d_out_ptr, C.byref(n)), "sdrx_fdecim_process_dev")
        return n.value

    def sync(self):
        _check(lib().sdrx_fdecim_sync(self._h), "sdrx_fdecim_sync")

    def set_stream(self, hip_stream: int | None):
        _check(lib().sdrx_fdecim_set_stream(self._h, hip_stream), "sdrx_fdecim_set_stream")

    def set_timing(self, on: bool):
        _check(lib().sdrx_fdecim_set_timing(self._h, int(on)), "sdrx_fdecim_set_timing")

    def get_timing(self, reset: bool = True):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().sdrx_fdecim_get_timing(self._h, C.byref(ms), C.byref(n), int(reset)), "sdrx_fdecim_get_timing")
        return ms.value, n.value

    def last_launch(self) -> dict:
        name = C.create_string_buffer(128)
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        _check(lib().sdrx_fdecim_last_launch(self._h, name, 128, C.byref(g), C.byref(b), C.byref(l)), "last_launch")
        return {"kernel": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}


class DcCorrection:
    """DSPDeviceSourceEngine::iqCorrections(begin, end, false): the DC offset correction of the device stream"""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().sdrx_dccorr_create(C.byref(self._h), device), "sdrx_dccorr_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_dccorr_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_dccorr_reset(self._h), "sdrx_dccorr_reset")

    def process(self, iq) -> np.ndarray:
        """returns the corrected copy of an int16 I/Q span (the C call works in place)"""
        buf = np.array(iq, dtype=np.int16, copy=True)
        _check(lib().sdrx_dccorr_process(self._h, buf.ctypes.data, buf.size // 2), "sdrx_dccorr_process")
        return buf

    def process_dev(self, d_in_ptr: int, d_out_ptr: int, n_cplx: int):
        _check(lib().sdrx_dccorr_process_dev(self._h, d_in_ptr, d_out_ptr, n_cplx), "sdrx_dccorr_process_dev")

    def sync(self):
        _check(lib().sdrx_dccorr_sync(self._h), "sdrx_dccorr_sync")

    def set_stream(self, hip_stream: int | None):
        _check(lib().sdrx_dccorr_set_stream(self._h, hip_stream), "sdrx_dccorr_set_stream")


def chan_plan(in_rate: int, req_rate: int, req_fc: int):
    """DownChannelizer::applyConfiguration's float bisection -> (modes, out_rate, residual_ofs)."""
    modes = np.zeros(32, np.uint8)
    r, f = C.c_int32(), C.c_int32()
    n = lib().sdrx_chan_plan(in_rate, req_rate, req_fc, modes.ctypes.data, C.byref(r), C.byref(f))
    if n < 0:
        raise SdrxError(f"sdrx_chan_plan rc={n}")
    return modes[:n].copy(), r.value, f.value


class ChannelizerBank:
    """N x DownChannelizer fed from one device stream (sdrbase/dsp/downchannelizer.{h,cpp})."""

    def __init__(self, in_rate: int, req_rates, req_fcs, device: int = 0):
        rr = np.ascontiguousarray(req_rates, dtype=np.int32)
        fc = np.ascontiguousarray(req_fcs, dtype=np.int32)
        assert rr.size == fc.size
        self.n_ch = int(rr.size)
        self._h = C.c_void_p()
        _check(lib().sdrx_chan_bank_create(C.byref(self._h), device, in_rate, self.n_ch, rr.ctypes.data, fc.ctypes.data),
               "sdrx_chan_bank_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_chan_bank_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def info(self, ch: int):
        n, r, f = C.c_int32(), C.c_int32(), C.c_int32()
        modes = np.zeros(32, np.uint8)
        _check(lib().sdrx_chan_bank_info(self._h, ch, C.byref(n), modes.ctypes.data, C.byref(r), C.byref(f)), "sdrx_chan_bank_info")
        return modes[: n.value].copy(), r.value, f.value

    def add_channel(self, req_rate: int, req_fc: int) -> int:
        c = C.c_int32(-1)
        _check(lib().sdrx_chan_bank_add_channel(self._h, req_rate, req_fc, C.byref(c)), "sdrx_chan_bank_add_channel")
        self.n_ch = max(getattr(self, "n_ch", 0), c.value + 1)
        return c.value

    def remove_channel(self, ch: int):
        _check(lib().sdrx_chan_bank_remove_channel(self._h, ch), "sdrx_chan_bank_remove_channel")

    @property
    def group_count(self) -> int:
        return lib().sdrx_chan_bank_group_count(self._h)

    def reconfigure(self, ch: int, req_rate: int, req_fc: int):
        _check(lib().sdrx_chan_bank_reconfigure(self._h, ch, req_rate, req_fc), "sdrx_chan_bank_reconfigure")

    def reset(self):
        _check(lib().sdrx_chan_bank_reset(self._h), "sdrx_chan_bank_reset")

    def feed(self, iq):
        iq = _i16(iq)
        _check(lib().sdrx_chan_bank_feed(self._h, iq.ctypes.data, iq.size // 2), "sdrx_chan_bank_feed")

    def feed_dev(self, d_ptr: int, n_cplx: int):
        _check(lib().sdrx_chan_bank_feed_dev(self._h, d_ptr, n_cplx), "sdrx_chan_bank_feed_dev")

    def available(self, ch: int) -> int:
        return lib().sdrx_chan_bank_available(self._h, ch)

    def read(self, ch: int, cap: int | None = None) -> np.ndarray:
        cap = self.available(ch) if cap is None else cap
        out = np.empty(max(2 * cap, 2), np.int16)
        n = lib().sdrx_chan_bank_read(self._h, ch, out.ctypes.data, cap)
        if n < 0:
            raise SdrxError(f"sdrx_chan_bank_read rc={n}: {lib().sdrx_last_error().decode()}")
        return out[: 2 * n]

    def last_dev(self, ch: int):
        """(device pointer, n_cplx) of what the last feed produced for channel ch"""
        p, n = C.c_void_p(), C.c_int64()
        _check(lib().sdrx_chan_bank_last_dev(self._h, ch, C.byref(p), C.byref(n)), "sdrx_chan_bank_last_dev")
        return p.value or 0, n.value

    def skip(self, ch: int, n: int = -1) -> int:
        return lib().sdrx_chan_bank_skip(self._h, ch, n)

    def sync(self):
        _check(lib().sdrx_chan_bank_sync(self._h), "sdrx_chan_bank_sync")

    def set_stream(self, hip_stream: int | None):
        _check(lib().sdrx_chan_bank_set_stream(self._h, hip_stream), "sdrx_chan_bank_set_stream")

    def set_timing(self, on: bool):
        _check(lib().sdrx_chan_bank_set_timing(self._h, int(on)), "sdrx_chan_bank_set_timing")

    def get_timing(self, reset: bool = True):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().sdrx_chan_bank_get_timing(self._h, C.byref(ms), C.byref(n), int(reset)), "sdrx_chan_bank_get_timing")
        return ms.value, n.value

    def last_launch(self) -> dict:
        name = C.create_string_buffer(128)
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        _check(lib().sdrx_chan_bank_last_launch(self._h, name, 128, C.byref(g), C.byref(b), C.byref(l)), "last_launch")
        return {"kernel": name.value.decode(), "grid": g.value, "block": b.value, "lds_bytes": l.value}


class FirCfg(C.Structure):
    """sdrx_fir_cfg"""
    _fields_ = [("kind", C.c_int32), ("ntaps", C.c_int32), ("sample_rate", C.c_float), ("f1", C.c_float), ("f2", C.c_float)]


class FirBank:
    """Lowpass<Real> / Bandpass<Real> (sdrbase/dsp/lowpass.h, bandpass.h) for N channels."""

    def __init__(self, cfgs, device: int = 0):
        self.n_ch = len(cfgs)
        arr = (FirCfg * self.n_ch)(*cfgs)
        self._h = C.c_void_p()
        _check(lib().sdrx_firbank_create(C.byref(self._h), device, self.n_ch, arr), "sdrx_firbank_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_firbank_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def taps(self, ch: int) -> np.ndarray:
        t = np.zeros(4096, np.float32)
        n = lib().sdrx_firbank_get_taps(self._h, ch, t.ctypes.data, t.size)
        return t[:n].copy()

    def feed(self, per_channel):
        ins = [np.ascontiguousarray(x, dtype=np.float32) for x in per_channel]
        outs = [np.empty(max(x.size, 1), np.float32) for x in ins]
        pi = (C.c_void_p * self.n_ch)(*[x.ctypes.data for x in ins])
        po = (C.c_void_p * self.n_ch)(*[x.ctypes.data for x in outs])
        ns = (C.c_int64 * self.n_ch)(*[x.size for x in ins])
        _check(lib().sdrx_firbank_feed(self._h, pi, ns, po), "sdrx_firbank_feed")
        return [o[: x.size] for o, x in zip(outs, ins)]


class SdriqHeader(C.Structure):
    """sdrx_sdriq_header (FileRecord::Header, sdrbase/dsp/filerecord.h:17-23)"""
    _fields_ = [("sample_rate", C.c_int32), ("center_frequency", C.c_uint64), ("start_timestamp", C.c_int64), ("sample_size", C.c_uint32)]


def sdriq_parse(data: bytes):
    """-> (SdriqHeader, int16 I/Q array of the samples that follow the 24-byte header)"""
    h = SdriqHeader()
    _check(lib().sdrx_sdriq_parse_header(data, len(data), C.byref(h)), "sdrx_sdriq_parse_header")
    body = np.frombuffer(data, dtype=np.int16, offset=24, count=(len(data) - 24) // 4 * 2) if h.sample_size == 16 else None
    return h, body


def sdriq_header_bytes(sample_rate: int, center_frequency: int, timestamp: int = 0, sample_size: int = 16) -> bytes:
    h = SdriqHeader(sample_rate, center_frequency, timestamp, sample_size)
    buf = C.create_string_buffer(24)
    _check(lib().sdrx_sdriq_write_header(buf, C.byref(h)), "sdrx_sdriq_write_header")
    return buf.raw


class BackendCfg(C.Structure):
    """sdrx_backend_cfg"""
    _fields_ = [("in_rate", C.c_int32), ("nco_freq", C.c_int32), ("out_rate", C.c_int32),
                ("interp_cutoff", C.c_float), ("taps_per_phase", C.c_float), ("filt_mode", C.c_int32),
                ("f1", C.c_float), ("f2", C.c_float), ("discri", C.c_int32), ("fm_scaling", C.c_float)]


class BackendBank:
    """NCO -> Interpolator -> fftfilt -> discriminator for N channels (front of the channelrx demods)."""

    def __init__(self, cfgs, device: int = 0):
        self.n_ch = len(cfgs)
        arr = (BackendCfg * self.n_ch)(*cfgs)
        self._h = C.c_void_p()
        _check(lib().sdrx_backend_create(C.byref(self._h), device, self.n_ch, arr), "sdrx_backend_create")
        self.cfgs = list(cfgs)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_backend_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def feed(self, per_channel_iq):
        bufs = [_i16(x) for x in per_channel_iq]
        ptrs = (C.c_void_p * self.n_ch)(*[b.ctypes.data for b in bufs])
        ns = (C.c_int64 * self.n_ch)(*[b.size // 2 for b in bufs])
        _check(lib().sdrx_backend_feed(self._h, ptrs, ns), "sdrx_backend_feed")

    def feed_bank(self, bank: "ChannelizerBank"):
        """channel c takes what the bank's last feed produced for its channel c; ordered on the device, no host sync"""
        _check(lib().sdrx_backend_feed_bank(self._h, bank._h), "sdrx_backend_feed_bank")

    def feed_dev(self, ptrs, counts):
        p = (C.c_void_p * self.n_ch)(*ptrs)
        n = (C.c_int64 * self.n_ch)(*counts)
        _check(lib().sdrx_backend_feed_dev(self._h, p, n), "sdrx_backend_feed_dev")

    def read(self, ch: int, cap_floats: int = 1 << 24) -> np.ndarray:
        out = np.empty(cap_floats, np.float32)
        n = lib().sdrx_backend_read(self._h, ch, out.ctypes.data, cap_floats)
        if n < 0:
            raise SdrxError(f"sdrx_backend_read rc={n}: {lib().sdrx_last_error().decode()}")
        return out[:n].copy()

    def sync(self):
        _check(lib().sdrx_backend_sync(self._h), "sdrx_backend_sync")

    def design(self, ch: int):
        nt, inc = C.c_int32(), C.c_int32()
        taps = np.zeros(16 * 256, np.float32)
        filt = np.zeros(4096, np.float32)
        _check(lib().sdrx_backend_get_design(self._h, ch, C.byref(nt), taps.ctypes.data, taps.size, filt.ctypes.data, C.byref(inc)),
               "sdrx_backend_get_design")
        return nt.value, taps[: 16 * nt.value].copy(), filt, inc.value


class AudioTailCfg(C.Structure):
    """sdrx_audiotail_cfg (include/sdrx.h)"""
    _fields_ = [("kind", C.c_int32), ("audio_rate", C.c_int32), ("volume", C.c_float),
                ("fm_scaling", C.c_float), ("squelch_level", C.c_float), ("squelch_gate", C.c_int32), ("af_bandwidth", C.c_float),
                ("agc_active", C.c_int32), ("agc_nb_samples", C.c_int32), ("agc_threshold_enable", C.c_int32), ("agc_gate", C.c_int32),
                ("agc_clamping", C.c_int32), ("agc_threshold", C.c_double)]


class AudioTail:
    """audio-rate tail of the NFM / SSB demods (squelch / MagAGC / delay line / Bandpass -> qint16) for N channels"""

    def __init__(self, cfgs, device: int = 0):
        self.n_ch = len(cfgs)
        arr = (AudioTailCfg * self.n_ch)(*cfgs)
        self._h = C.c_void_p()
        _check(lib().sdrx_audiotail_create(C.byref(self._h), device, self.n_ch, arr), "sdrx_audiotail_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_audiotail_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_audiotail_reset(self._h), "sdrx_audiotail_reset")

    def feed(self, per_channel_cplx):
        ins = [np.ascontiguousarray(x, dtype=np.float32) for x in per_channel_cplx]
        outs = [np.zeros(max(x.size // 2, 1), np.int16) for x in ins]
        pi = (C.c_void_p * self.n_ch)(*[x.ctypes.data for x in ins])
        po = (C.c_void_p * self.n_ch)(*[x.ctypes.data for x in outs])
        ns = (C.c_int64 * self.n_ch)(*[x.size // 2 for x in ins])
        _check(lib().sdrx_audiotail_feed(self._h, pi, ns, po), "sdrx_audiotail_feed")
        return [o[: x.size // 2] for o, x in zip(outs, ins)]


class IirCfg(C.Structure):
    _fields_ = [("order", C.c_int32), ("a", C.c_float * 9), ("b", C.c_float * 9)]


class IirBank:
    """IIRFilter<float, Order> (sdrbase/dsp/iirfilter.h), one filter per channel"""

    def __init__(self, specs, device: int = 0):
        """specs: list of (order, a, b)"""
        self.n_ch = len(specs)
        arr = (IirCfg * self.n_ch)()
        for i, (o, a, b) in enumerate(specs):
            arr[i].order = o
            for j in range(o + 1):
                arr[i].a[j] = a[j]; arr[i].b[j] = b[j]
        self._h = C.c_void_p()
        _check(lib().sdrx_iir_create(C.byref(self._h), device, self.n_ch, arr), "sdrx_iir_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_iir_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_iir_reset(self._h), "sdrx_iir_reset")

    def feed(self, per_channel):
        ins = [np.ascontiguousarray(x, dtype=np.float32) for x in per_channel]
        outs = [np.zeros(max(x.size, 1), np.float32) for x in ins]
        pi = (C.c_void_p * self.n_ch)(*[x.ctypes.data for x in ins])
        po = (C.c_void_p * self.n_ch)(*[x.ctypes.data for x in outs])
        ns = (C.c_int64 * self.n_ch)(*[x.size for x in ins])
        _check(lib().sdrx_iir_feed(self._h, pi, ns, po), "sdrx_iir_feed")
        return [o[: x.size] for o, x in zip(outs, ins)]


class Decimators24:
    """Decimators<qint32, qint16, 24, InputBits> of the reference's 24-bit sample build (decimators.h, SDR_RX_SAMPLE_24BIT):
    int16 I/Q in, {int32, int32} samples out; same call contract as Decimators.decimate()."""

    def __init__(self, log2_decim: int, fcpos: int = FC_CEN, input_bits: int = 12, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().sdrx_decim24_create(C.byref(self._h), device, log2_decim, fcpos, input_bits), "sdrx_decim24_create")
        self.log2 = log2_decim

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_decim24_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_decim24_reset(self._h), "sdrx_decim24_reset")

    def decimate(self, buf) -> np.ndarray:
        buf = np.ascontiguousarray(buf, dtype=np.int16)
        out = np.empty(2 * ((buf.size // 2) >> self.log2) + 2, np.int32)
        n = C.c_int32()
        _check(lib().sdrx_decim24_process(self._h, buf.ctypes.data, buf.size, out.ctypes.data, C.byref(n)), "sdrx_decim24_process")
        return out[: 2 * n.value]


    def decimate_dev(self, d_iq: int, n_cplx: int, d_out: int) -> int:
        """device pointers (e.g. torch tensor .data_ptr()); asynchronous, sync() waits; returns the samples produced"""
        n = C.c_int64()
        _check(lib().sdrx_decim24_process_dev(self._h, d_iq, n_cplx, d_out, C.byref(n)), "sdrx_decim24_process_dev")
        return n.value

    def sync(self):
        _check(lib().sdrx_decim24_sync(self._h), "sdrx_decim24_sync")


class ChannelizerBank24:
    """N DownChannelizers (downchannelizer.cpp) of the 24-bit sample build on one {int32, int32} stream."""

    def __init__(self, in_rate: int, req_rates, req_fcs, device: int = 0):
        rr = np.ascontiguousarray(req_rates, dtype=np.int32); rf = np.ascontiguousarray(req_fcs, dtype=np.int32)
        self.n_ch = rr.size
        self._h = C.c_void_p()
        _check(lib().sdrx_chan24_bank_create(C.byref(self._h), device, in_rate, self.n_ch, rr.ctypes.data, rf.ctypes.data), "sdrx_chan24_bank_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_chan24_bank_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_chan24_bank_reset(self._h), "sdrx_chan24_bank_reset")

    def info(self, ch: int):
        n, r, f = C.c_int32(), C.c_int32(), C.c_int32()
        modes = np.zeros(40, np.uint8)
        _check(lib().sdrx_chan24_bank_info(self._h, ch, C.byref(n), modes.ctypes.data, C.byref(r), C.byref(f)), "sdrx_chan24_bank_info")
        return modes[: n.value].copy(), r.value, f.value

    def feed_dev(self, d_iq: int, n_cplx: int):
        _check(lib().sdrx_chan24_bank_feed_dev(self._h, d_iq, n_cplx), "sdrx_chan24_bank_feed_dev")

    def out_dev(self, ch: int):
        p, n = C.c_void_p(), C.c_int64()
        _check(lib().sdrx_chan24_bank_out_dev(self._h, ch, C.byref(p), C.byref(n)), "sdrx_chan24_bank_out_dev")
        return p.value, n.value

    def sync(self):
        _check(lib().sdrx_chan24_bank_sync(self._h), "sdrx_chan24_bank_sync")

    def feed(self, iq):
        """iq: interleaved int32 I/Q; returns the per-channel outputs of this feed"""
        iq = np.ascontiguousarray(iq, dtype=np.int32)
        _check(lib().sdrx_chan24_bank_feed(self._h, iq.ctypes.data, iq.size // 2), "sdrx_chan24_bank_feed")
        outs = []
        for c in range(self.n_ch):
            out = np.empty(iq.size + 2, np.int32)
            n = lib().sdrx_chan24_bank_read(self._h, c, out.ctypes.data, out.size // 2)
            if n < 0:
                _check(int(n), "sdrx_chan24_bank_read")
            outs.append(out[: 2 * n].copy())
        return outs


class IqImbalance:
    """DSPDeviceSourceEngine::iqCorrections(begin, end, true) (DC + I/Q imbalance, float flavour) for N device streams."""

    def __init__(self, n_streams: int, device: int = 0):
        self.n = n_streams
        self._h = C.c_void_p()
        _check(lib().sdrx_iqimb_create(C.byref(self._h), device, n_streams), "sdrx_iqimb_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_iqimb_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def reset(self):
        _check(lib().sdrx_iqimb_reset(self._h), "sdrx_iqimb_reset")

    def process(self, per_stream_iq):
        """in place on copies: returns the corrected int16 I/Q per stream"""
        bufs = [_i16(x).copy() for x in per_stream_iq]
        ptrs = (C.c_void_p * self.n)(*[b.ctypes.data for b in bufs])
        ns = (C.c_int64 * self.n)(*[b.size // 2 for b in bufs])
        _check(lib().sdrx_iqimb_process(self._h, ptrs, ns), "sdrx_iqimb_process")
        return bufs


class SampleSinkFifo:
    """sdrbase/dsp/samplesinkfifo.{h,cpp}: write / read / readBegin / readCommit."""

    def __init__(self, size: int):
        self._h = C.c_void_p()
        _check(lib().sdrx_fifo_create(C.byref(self._h), size), "sdrx_fifo_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdrx_fifo_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = _del

    def set_size(self, size: int):
        _check(lib().sdrx_fifo_set_size(self._h, size), "sdrx_fifo_set_size")

    size = property(lambda self: lib().sdrx_fifo_size(self._h))
    fill = property(lambda self: lib().sdrx_fifo_fill(self._h))
    dropped = property(lambda self: lib().sdrx_fifo_dropped(self._h))

    def write(self, iq) -> int:
        iq = _i16(iq)
        return lib().sdrx_fifo_write(self._h, iq.ctypes.data, iq.size // 2)

    def write_bytes(self, data: bytes) -> int:
        return lib().sdrx_fifo_write_bytes(self._h, data, len(data))

    def read(self, count: int) -> np.ndarray:
        out = np.empty(max(2 * count, 2), np.int16)
        n = lib().sdrx_fifo_read(self._h, out.ctypes.data, count)
        return out[: 2 * n]

    def read_begin(self, count: int):
        p1, p2, n1, n2 = C.c_void_p(), C.c_void_p(), C.c_uint32(), C.c_uint32()
        tot = lib().sdrx_fifo_read_begin(self._h, count, C.byref(p1), C.byref(n1), C.byref(p2), C.byref(n2))

        def view(p, n):
            if not n:
                return np.empty(0, np.int16)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(2 * n,)).copy()

        return tot, view(p1, n1.value), view(p2, n2.value)

    def read_commit(self, count: int) -> int:
        return lib().sdrx_fifo_read_commit(self._h, count)


def measure_hbm_read(device: int = 0, n_bytes: int = 4 << 30, reps: int = 5) -> float:
    """GB/s of a read-only streaming kernel over n_bytes of HBM (best of reps) -- the measured roofline denominator"""
    v = C.c_double()
    _check(lib().sdrx_measure_hbm_read(device, n_bytes, reps, C.byref(v)), "sdrx_measure_hbm_read")
    return v.value
