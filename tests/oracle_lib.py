"""ctypes loaders for the TEST ORACLES (never imported by the product):

* ``Oracle``  - oracle/libhsrans_oracle.so, the plain-C restatement of the reference algorithms;
* ``Ref``     - oracle/_ref/libhsrans_ref.so, the real reference compiled from /root/reference/src (may be absent).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libhsrans_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libhsrans_ref.so")

RAW, BLOCK, MT = 0, 1, 2
CONTAINER_NAMES = {RAW: "raw", BLOCK: "block", MT: "mt"}

_sz = ctypes.c_size_t
_vp = ctypes.c_void_p
_i = ctypes.c_int
_u = ctypes.c_uint


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)


class OrcHist(ctypes.Structure):
    _fields_ = [("symbolCount", ctypes.c_uint16 * 256), ("cumul", ctypes.c_uint16 * 256)]


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "hsrans_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        L = ctypes.CDLL(ORACLE_SO)
        L.orc_idx2idx.restype = ctypes.c_uint8
        L.orc_idx2idx.argtypes = [_u]
        L.orc_make_hist.restype = None
        L.orc_make_hist.argtypes = [ctypes.POINTER(OrcHist), _vp, _sz, _u]
        L.orc_make_dec_table.restype = _i
        L.orc_make_dec_table.argtypes = [_u, _vp, _vp, _vp, _i]
        L.orc_capacity.restype = _sz
        L.orc_capacity.argtypes = [_i, _i, _sz]
        L.orc_raw_encode.restype = _sz
        L.orc_raw_encode.argtypes = [_i, _u, _vp, _sz, _vp, _sz, ctypes.POINTER(OrcHist)]
        L.orc_decode.restype = _sz
        L.orc_decode.argtypes = [_i, _i, _u, _vp, _sz, _vp, _sz]
        L.orc_exec_plan.restype = _sz
        L.orc_exec_plan.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz]
        self.L = L

    def exec_plan(self, plan: np.ndarray, stream: np.ndarray, out_cap: int):
        plan = np.ascontiguousarray(plan, dtype=np.uint8)
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.full(max(out_cap, 1), 0xCC, np.uint8)
        r = self.L.orc_exec_plan(_ptr(plan), plan.size, _ptr(stream), stream.size, _ptr(out), out_cap)
        return r, out[:out_cap]

    def idx2idx(self, j):
        return self.L.orc_idx2idx(j)

    def capacity(self, container, states, n):
        return self.L.orc_capacity(container, states, n)

    def make_hist(self, data: np.ndarray, bits: int) -> OrcHist:
        h = OrcHist()
        self.L.orc_make_hist(ctypes.byref(h), _ptr(data), data.size, bits)
        return h

    def hist_from_counts(self, counts) -> OrcHist:
        h = OrcHist()
        c = 0
        for k in range(256):
            h.symbolCount[k] = int(counts[k])
            h.cumul[k] = c & 0xFFFF
            c += int(counts[k])
        return h

    def make_dec_table(self, bits, counts: np.ndarray, wide_sum=0):
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        cumul = np.zeros(256, np.uint16)
        inv = np.zeros(1 << bits, np.uint8)
        ok = self.L.orc_make_dec_table(bits, _ptr(counts), _ptr(cumul), _ptr(inv), wide_sum)
        return ok, cumul, inv

    def raw_encode(self, states, bits, data: np.ndarray, hist: OrcHist | None = None) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if hist is None:
            hist = self.make_hist(data, bits)
        cap = self.capacity(RAW, states, data.size)
        out = np.zeros(cap, np.uint8)
        m = self.L.orc_raw_encode(states, bits, _ptr(data), data.size, _ptr(out), cap, ctypes.byref(hist))
        return out[:m].copy()

    def decode(self, container, states, bits, stream: np.ndarray, out_cap: int, in_len: int | None = None):
        """Returns (returned_length, output buffer of out_cap bytes pre-filled with 0xCC)."""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.full(max(out_cap, 1), 0xCC, np.uint8)
        r = self.L.orc_decode(container, states, bits, _ptr(stream), stream.size if in_len is None else in_len, _ptr(out), out_cap)
        return r, out[:out_cap]


class Ref:
    """The real reference (only where oracle/_ref/libhsrans_ref.so has been built)."""

    def timed_decode(self, container, states, bits, stream: np.ndarray, out_cap: int, variant: int = 0, threads: int = 0, runs: int = 3, budget_s: float = 4.0,
                     max_runs: int = 40):
        """Best wall time of the C call alone (buffers allocated and touched beforehand), the output of the last run, and the run count."""
        import time

        buf = np.zeros(stream.size + 64, np.uint8)
        buf[:stream.size] = stream
        out = np.full(max(out_cap, 1) + 64, 0xCC, np.uint8)
        best, n, total = None, 0, 0.0
        while n < runs or (total < budget_s and n < max_runs):
            t0 = time.perf_counter()
            r = self.L.hsref_decode(container, states, bits, variant, _ptr(buf), stream.size, _ptr(out), out_cap, threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            total += dt
            n += 1
        return best, r, out[:out_cap], n

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def __init__(self):
        L = ctypes.CDLL(REF_SO)
        L.hsref_capacity.restype = _sz
        L.hsref_capacity.argtypes = [_i, _i, _sz]
        L.hsref_make_hist.restype = None
        L.hsref_make_hist.argtypes = [_vp, _sz, _i, _vp, _vp]
        L.hsref_encode.restype = _sz
        L.hsref_encode.argtypes = [_i, _i, _i, _vp, _sz, _vp, _sz, _vp]
        L.hsref_decode.restype = _sz
        L.hsref_decode.argtypes = [_i, _i, _i, _i, _vp, _sz, _vp, _sz, _i]
        L.hsref_pool_threads.restype = _sz
        L.hsref_has_avx2.restype = _i
        L.hsref_has_avx512.restype = _i
        self.L = L

    def has_avx512(self) -> bool:
        return bool(self.L.hsref_has_avx512())

    def capacity(self, container, states, n):
        return self.L.hsref_capacity(container, states, n)

    def make_hist(self, data: np.ndarray, bits: int):
        counts = np.zeros(256, np.uint16)
        cumul = np.zeros(256, np.uint16)
        self.L.hsref_make_hist(_ptr(data), data.size, bits, _ptr(counts), _ptr(cumul))
        return counts, cumul

    def encode(self, container, states, bits, data: np.ndarray, counts=None) -> np.ndarray:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        cap = self.capacity(container, states, data.size)
        out = np.zeros(cap + 64, np.uint8)
        cp = None
        if counts is not None:
            counts = np.ascontiguousarray(counts, dtype=np.uint16)
            cp = _ptr(counts)
        m = self.L.hsref_encode(container, states, bits, _ptr(data), data.size, _ptr(out), cap, cp)
        return out[:m].copy()

    def decode(self, container, states, bits, stream: np.ndarray, out_cap: int, variant: int = 0, threads: int = 0, pad: int = 64):
        """variant 0: scalar (raw) / runtime dispatch (block_, mt_); 1: fastest AVX2 raw; 2: mt_ thread pool; 3: fastest AVX-512 raw."""
        buf = np.zeros(stream.size + pad, np.uint8)  # SIMD variants over-read up to 32 B
        buf[:stream.size] = stream
        out = np.full(max(out_cap, 1) + 64, 0xCC, np.uint8)
        r = self.L.hsref_decode(container, states, bits, variant, _ptr(buf), stream.size, _ptr(out), out_cap, threads)
        return r, out[:out_cap]
