"""ctypes binding of include/hsrans_hip.h.

Mirrors the reference's codec interface (src/main.cpp:146-155): ``encode(bytes) -> stream`` and
``decode(stream) -> bytes`` selected by container (raw / block_ / mt_), state count (32 / 64) and histogram bits
(10..15) — the three things the reference encodes in its function names
(``rANS32x64_16w_decode_scalar_11``, ``mt_rANS32x32_16w_decode_14`` …).  Failure (the reference's ``return 0``) raises
``HsransError``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch  # first: pins the HIP runtime (libamdhip64.so.7) this process uses before our library is loaded

RAW, BLOCK, MT = 0, 1, 2
_HERE = os.path.dirname(os.path.abspath(__file__))
_sz, _vp, _i, _u32 = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32


class HsransError(RuntimeError):
    pass


class Hist(ctypes.Structure):
    """reference hist_t (src/hist.h:16-20)"""
    _fields_ = [("symbolCount", ctypes.c_uint16 * 256), ("cumul", ctypes.c_uint16 * 256)]


class EncodeOpts(ctypes.Structure):
    _fields_ = [("block_size", _u32), ("index_interval", _u32), ("plan_out", _vp), ("plan_capacity", _sz), ("plan_size", _sz), ("flags", _u32),
                ("reserved", _u32), ("index_groups", _vp), ("n_index_groups", _sz)]


class Calibration(ctypes.Structure):
    _fields_ = [("class_weights", _u32 * 8), ("class_finish_us_last_iteration", ctypes.c_double * 8), ("last_wave_us_before", ctypes.c_double),
                ("last_wave_us_after", ctypes.c_double), ("class_spread_us_before", ctypes.c_double), ("class_spread_us_after", ctypes.c_double),
                ("iterations", _u32), ("reserved", _u32), ("bytes", ctypes.c_uint64)]


class LaunchInfo(ctypes.Structure):
    _fields_ = [(n, _u32) for n in ("grid", "block", "lds_bytes", "waves_per_block", "chains", "shared_table", "walk", "two_level", "table_mode", "chains_per_wave")] + \
               [("class_weights", _u32 * 8), ("dynamic_groups", _u32), ("spread", _u32)]


class BatchInfo(ctypes.Structure):
    _fields_ = [(n, _u32) for n in ("members", "launches", "direct_members", "solo_members", "grouped_members", "reserved", "grid", "block", "lds_bytes")] + \
               [("class_weights", _u32 * 8), ("imbalance", ctypes.c_double)]


class QueueStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("submitted", "flushes", "launches", "batches_made", "batches_reused", "retargeted")]


class Shard(ctypes.Structure):
    _fields_ = [("first_chain", _u32), ("chain_count", _u32), ("out_begin", ctypes.c_uint64), ("out_end", ctypes.c_uint64)]


class ShardedInfo(ctypes.Structure):
    _fields_ = [("world", _u32), ("rank", _u32), ("parts", _u32), ("root", ctypes.c_int32), ("window_begin", ctypes.c_uint64), ("window_end", ctypes.c_uint64),
                ("out_base", ctypes.c_uint64), ("out_length", ctypes.c_uint64), ("decoded_length", ctypes.c_uint64), ("stream_length", ctypes.c_uint64),
                ("one_launch", _u32), ("reserved", _u32)]


COMM_ID_BYTES = 128
SHARD_DECODE_ONLY, SHARD_DECODE_AND_EXCHANGE, SHARD_EXCHANGE_ONLY = 0, 1, 2


def lib_path() -> str:
    # HSRANS_LIB: A/B a differently built libhsrans_hip.so in one process environment (kernel tuning only)
    if os.environ.get("HSRANS_LIB"):
        return os.environ["HSRANS_LIB"]
    if os.environ.get("HSRANS_DEBUG_STAMPS"):  # the per-wave stamps only exist in the diagnostic build (make -C csrc stamps)
        stamps = os.path.join(_HERE, "lib", "libhsrans_hip_stamps.so")
        if not os.path.exists(stamps):
            raise HsransError("HSRANS_DEBUG_STAMPS needs the diagnostic library: make -C hypersonic_rans_amd/csrc stamps")
        main = os.path.join(_HERE, "lib", "libhsrans_hip.so")
        # (`make` alone does not rebuild it: numbers from a diagnostic library older than the product one are numbers of old code)
        if os.path.exists(main) and os.path.getmtime(stamps) + 1 < os.path.getmtime(main):
            raise HsransError("libhsrans_hip_stamps.so is older than libhsrans_hip.so: make -C hypersonic_rans_amd/csrc all stamps")
        return stamps
    return os.path.join(_HERE, "lib", "libhsrans_hip.so")


_LIB = None


def load_library() -> ctypes.CDLL:
    """Loads lib/libhsrans_hip.so.  Fails loudly when it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` (there is no fallback implementation)")
    L = ctypes.CDLL(path)
    L.hsrans_version.restype = ctypes.c_char_p
    L.hsrans_capacity.restype = _sz
    L.hsrans_capacity.argtypes = [_i, _i, _sz]
    L.hsrans_make_hist.restype = None
    L.hsrans_make_hist.argtypes = [ctypes.POINTER(Hist), _vp, _sz, _u32]
    L.hsrans_encode.restype = _sz
    L.hsrans_encode.argtypes = [_i, _i, _u32, _vp, _sz, _vp, _sz, ctypes.POINTER(Hist)]
    L.hsrans_encode_ex.restype = _sz
    L.hsrans_encode_ex.argtypes = [_i, _i, _u32, _vp, _sz, _vp, _sz, ctypes.POINTER(Hist), ctypes.POINTER(EncodeOpts)]
    L.hsrans_plan_capacity.restype = _sz
    L.hsrans_plan_capacity.argtypes = [_i, _i, _sz, _u32, _u32]
    L.hsrans_plan_build.restype = _sz
    L.hsrans_plan_build.argtypes = [_i, _i, _u32, _vp, _sz, _sz, _vp, _sz]
    L.hsrans_plan_chain_count.restype = _u32
    L.hsrans_plan_chain_count.argtypes = [_vp, _sz]
    L.hsrans_plan_decoded_length.restype = ctypes.c_uint64
    L.hsrans_plan_decoded_length.argtypes = [_vp, _sz]
    L.hsrans_plan_slice.restype = _sz
    L.hsrans_plan_slice.argtypes = [_vp, _sz, _u32, _u32, _vp, _sz]
    L.hsrans_plan_stream_ranges.restype = _i
    L.hsrans_plan_stream_ranges.argtypes = [_vp, _sz, _u32, _u32, ctypes.POINTER(ctypes.c_uint64)]
    L.hsrans_plan_chain_range.restype = _i
    L.hsrans_plan_chain_range.argtypes = [_vp, _sz, _u32, _u32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.hsrans_cpu_level.restype = _i
    L.hsrans_decode_cpu.restype = _sz
    L.hsrans_decode_cpu.argtypes = [_i, _u32, _i, _i, _u32, _vp, _sz, _vp, _sz, _vp, _sz]
    L.hsrans_index_build_host.restype = _sz
    L.hsrans_index_build_host.argtypes = [_i, _u32, _i, _i, _u32, _vp, _sz, _vp, _sz, _vp, _sz]
    L.hsrans_ctx_create.restype = _i
    L.hsrans_ctx_create.argtypes = [_i, ctypes.POINTER(_vp)]
    L.hsrans_ctx_destroy.restype = None
    L.hsrans_ctx_destroy.argtypes = [_vp]
    L.hsrans_ctx_host_index_chains.restype = _u32
    L.hsrans_ctx_host_index_chains.argtypes = [_vp]
    L.hsrans_ctx_device_name.restype = ctypes.c_char_p
    L.hsrans_ctx_device_name.argtypes = [_vp]
    L.hsrans_decode_host.restype = _sz
    L.hsrans_decode_host.argtypes = [_vp, _i, _i, _u32, _vp, _sz, _vp, _sz, _vp, _sz]
    L.hsrans_dplan_create.restype = _i
    L.hsrans_dplan_create.argtypes = [_vp, _vp, _sz, ctypes.POINTER(_vp)]
    L.hsrans_dplan_destroy.restype = None
    L.hsrans_dplan_destroy.argtypes = [_vp]
    L.hsrans_decode_device.restype = _i
    L.hsrans_decode_device.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp]
    L.hsrans_decode_device_window.restype = _i
    L.hsrans_decode_device_window.argtypes = [_vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp]
    L.hsrans_decode_device_ranges.restype = _i
    L.hsrans_decode_device_ranges.argtypes = [_vp, _vp, _vp, _sz, _sz, _vp, _sz, _sz, _vp]
    L.hsrans_decode_device_indexing.restype = _i
    L.hsrans_decode_device_indexing.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _u32, _vp, ctypes.POINTER(_vp)]
    L.hsrans_ctx_calibrate.restype = _i
    L.hsrans_ctx_calibrate.argtypes = [_vp, _u32, _u32, ctypes.POINTER(Calibration)]
    L.hsrans_dplan_batch_create.restype = _i
    L.hsrans_dplan_batch_create.argtypes = [_vp, _vp, _u32, ctypes.POINTER(_vp)]
    L.hsrans_dplan_batch_destroy.restype = None
    L.hsrans_dplan_batch_destroy.argtypes = [_vp]
    L.hsrans_decode_device_batch.restype = _i
    L.hsrans_decode_device_batch.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.hsrans_dplan_batch_status.restype = _i
    L.hsrans_dplan_batch_status.argtypes = [_vp, _vp, _vp, _vp]
    L.hsrans_dplan_batch_info.restype = _i
    L.hsrans_dplan_batch_info.argtypes = [_vp, ctypes.POINTER(BatchInfo)]
    L.hsrans_queue_create.restype = _i
    L.hsrans_queue_create.argtypes = [_vp, _u32, ctypes.POINTER(_vp)]
    L.hsrans_queue_destroy.restype = None
    L.hsrans_queue_destroy.argtypes = [_vp]
    L.hsrans_queue_submit.restype = _i
    L.hsrans_queue_submit.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp]
    L.hsrans_queue_flush.restype = _i
    L.hsrans_queue_flush.argtypes = [_vp, _vp]
    L.hsrans_queue_pending.restype = _u32
    L.hsrans_queue_pending.argtypes = [_vp]
    L.hsrans_queue_stats.restype = _i
    L.hsrans_queue_stats.argtypes = [_vp, ctypes.POINTER(QueueStats)]
    L.hsrans_dealt_shares.restype = _i
    L.hsrans_dealt_shares.argtypes = [_vp, _u32, _vp, _u32, _u32, ctypes.c_uint64, _vp, _vp]
    L.hsrans_batch_deal.restype = ctypes.c_double
    L.hsrans_batch_deal.argtypes = [_vp, _vp, _u32, _u32, _u32, _vp, _vp]
    L.hsrans_dplan_batch_read_finish.restype = _sz
    L.hsrans_dplan_batch_read_finish.argtypes = [_vp, _vp, _sz]
    L.hsrans_ctx_calibrate_runs.restype = _i
    L.hsrans_ctx_calibrate_runs.argtypes = [_vp, _u32, _u32, _u32, ctypes.POINTER(Calibration)]
    L.hsrans_index_boundaries_batch.restype = _sz
    L.hsrans_index_boundaries_batch.argtypes = [_vp, _i, _u32, _vp, _u32, _u32, _vp, _sz]
    L.hsrans_comm_unique_id.restype = _i
    L.hsrans_comm_unique_id.argtypes = [_vp]
    L.hsrans_comm_create.restype = _i
    L.hsrans_comm_create.argtypes = [_vp, _vp, _i, _i, ctypes.POINTER(_vp)]
    L.hsrans_comm_destroy.restype = None
    L.hsrans_comm_destroy.argtypes = [_vp]
    L.hsrans_comm_rank.restype = _i
    L.hsrans_comm_rank.argtypes = [_vp]
    L.hsrans_comm_world.restype = _i
    L.hsrans_comm_world.argtypes = [_vp]
    L.hsrans_comm_rccl_version.restype = _i
    L.hsrans_shard_layout.restype = _i
    L.hsrans_shard_layout.argtypes = [_vp, _sz, _u32, _u32, _vp, _vp, _vp]
    L.hsrans_sharded_create.restype = _i
    L.hsrans_sharded_create.argtypes = [_vp, _vp, _vp, _sz, _u32, _vp, _i, ctypes.POINTER(_vp)]
    L.hsrans_sharded_create_rank.restype = _i
    L.hsrans_sharded_create_rank.argtypes = [_vp, _i, _i, _vp, _sz, _u32, _vp, _i, ctypes.POINTER(_vp)]
    L.hsrans_sharded_destroy.restype = None
    L.hsrans_sharded_destroy.argtypes = [_vp]
    L.hsrans_sharded_info.restype = _i
    L.hsrans_sharded_info.argtypes = [_vp, ctypes.POINTER(ShardedInfo), _vp, _sz]
    L.hsrans_sharded_part_plan.restype = _vp
    L.hsrans_sharded_part_plan.argtypes = [_vp, _u32]
    L.hsrans_sharded_wait_part.restype = _i
    L.hsrans_sharded_wait_part.argtypes = [_vp, _u32, _vp]
    L.hsrans_sharded_whole_plan.restype = _vp
    L.hsrans_sharded_whole_plan.argtypes = [_vp]
    L.hsrans_decode_sharded.restype = _i
    L.hsrans_decode_sharded.argtypes = [_vp, _vp, _vp, _i, _vp]
    L.hsrans_sharded_status.restype = _i
    L.hsrans_sharded_status.argtypes = [_vp, _vp]
    L.hsrans_dplan_status.restype = _i
    L.hsrans_dplan_status.argtypes = [_vp, _vp, _vp]
    L.hsrans_dplan_launch_info.restype = _i
    L.hsrans_dplan_launch_info.argtypes = [_vp, ctypes.POINTER(LaunchInfo)]
    L.hsrans_dplan_create_from_device_stream.restype = _i
    L.hsrans_dplan_create_from_device_stream.argtypes = [_vp, _i, _i, _u32, _vp, _sz, _sz, _vp, ctypes.POINTER(_vp)]
    L.hsrans_dplan_read_plan.restype = _sz
    L.hsrans_dplan_read_plan.argtypes = [_vp, _vp, _sz]
    L.hsrans_encode_device_raw.restype = _sz
    L.hsrans_encode_device_raw.argtypes = [_vp, _i, _u32, _vp, _sz, _vp, _sz, _vp, _u32, _vp, _sz, _vp, _sz, ctypes.POINTER(_sz), _vp, ctypes.POINTER(_vp)]
    L.hsrans_encode_device.restype = _sz
    L.hsrans_encode_device.argtypes = [_vp, _i, _i, _u32, _vp, _sz, _vp, _sz, _u32, _u32, _vp, ctypes.POINTER(_vp)]
    L.hsrans_index_build.restype = _sz
    L.hsrans_index_build.argtypes = [_vp, _i, _i, _u32, _vp, _sz, _u32, _vp, _sz]
    L.hsrans_index_build_at.restype = _sz
    L.hsrans_index_build_at.argtypes = [_vp, _i, _i, _u32, _vp, _sz, _vp, _sz, _vp, _sz]
    L.hsrans_index_boundaries.restype = _sz
    L.hsrans_index_boundaries.argtypes = [_vp, _i, _u32, _sz, _vp, _sz]
    L.hsrans_plan_thin.restype = _sz
    L.hsrans_plan_thin.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz]
    L.hsrans_plan_capacity_chains.restype = _sz
    L.hsrans_plan_capacity_chains.argtypes = [_i, _i, _sz, _sz, _u32]
    L.hsrans_hpipe_create.restype = _i
    L.hsrans_hpipe_create.argtypes = [_vp, _vp, _sz, _u32, ctypes.POINTER(_vp)]
    L.hsrans_hpipe_decode.restype = _sz
    L.hsrans_hpipe_decode.argtypes = [_vp, _vp, _sz, _vp, _sz]
    L.hsrans_hpipe_destroy.restype = None
    L.hsrans_hpipe_destroy.argtypes = [_vp]
    L.hsrans_decode_host_pipelined.restype = _sz
    L.hsrans_decode_host_pipelined.argtypes = [_vp, _i, _i, _u32, _vp, _sz, _vp, _sz, _vp, _sz, _u32]
    L.hsrans_host_register.restype = _i
    L.hsrans_host_register.argtypes = [_vp, _vp, _sz]
    L.hsrans_host_unregister.restype = _i
    L.hsrans_host_unregister.argtypes = [_vp, _vp]
    _LIB = L
    return L


def _u8(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8) if not isinstance(a, (bytes, bytearray, memoryview)) else np.frombuffer(a, dtype=np.uint8)
    return a


def _p(a: np.ndarray):
    return a.ctypes.data_as(_vp)


# ---------------------------------------------------------------------------------------------------------------------
# host-side format functions
# ---------------------------------------------------------------------------------------------------------------------
def capacity(container: int, states: int, n: int) -> int:
    return load_library().hsrans_capacity(container, states, n)


def make_hist(data, bits: int) -> Hist:
    data = _u8(data)
    h = Hist()
    load_library().hsrans_make_hist(ctypes.byref(h), _p(data), data.size, bits)
    return h


def hist_from_counts(counts) -> Hist:
    h = Hist()
    run = 0
    for k in range(256):
        h.symbolCount[k] = int(counts[k])
        h.cumul[k] = run & 0xFFFF
        run += int(counts[k])
    return h


ENC_INDEPENDENT_BLOCKS = 1


def index_boundaries(states: int, bits: int, decoded_size: int, ctx: "Context | None" = None) -> np.ndarray:
    """Checkpoint positions (group indices) for ONE chain per resident wavefront (hsrans_index_boundaries); ctx None = MI355X defaults."""
    out = np.zeros(1 << 16, np.uint64)
    n = load_library().hsrans_index_boundaries(ctx.handle if ctx is not None else None, states, bits, decoded_size, _p(out), out.size)
    return out[:n].copy()


def encode(container: int, states: int, bits: int, data, hist: Hist | None = None, block_size: int = 0, index_interval: int = 0,
           independent_blocks: bool = False, index_groups=None, out_capacity: int | None = None):
    """Returns ``stream`` (np.uint8) or ``(stream, plan)`` when ``index_interval`` != 0 or ``index_groups`` (explicit checkpoints) is given.
    ``out_capacity``: room for the stream when a caller-made histogram makes the data EXPAND (default: the reference's capacity)."""
    L = load_library()
    data = _u8(data)
    cap = L.hsrans_capacity(container, states, data.size) if out_capacity is None else int(out_capacity)
    out = np.zeros(cap, np.uint8)
    hp = ctypes.byref(hist) if hist is not None else None
    if index_groups is not None:
        index_groups = np.ascontiguousarray(index_groups, dtype=np.uint64)
        if index_groups.size == 0:  # a stream too short for more than one chain
            index_groups, index_interval = None, 4
    want_plan = index_interval != 0 or index_groups is not None
    if not want_plan and block_size == 0 and not independent_blocks:
        m = L.hsrans_encode(container, states, bits, _p(data), data.size, _p(out), cap, hp)
        if m == 0:
            raise HsransError("encode failed")
        return out[:m].copy()
    if index_groups is not None:
        pcap = L.hsrans_plan_capacity_chains(container, states, data.size, index_groups.size, block_size)
    else:
        pcap = L.hsrans_plan_capacity(container, states, data.size, index_interval, block_size) if index_interval else 0
    plan = np.zeros(max(pcap, 1), np.uint8)
    opts = EncodeOpts(block_size, 0 if index_groups is not None else index_interval, plan.ctypes.data if want_plan else None, pcap, 0,
                      ENC_INDEPENDENT_BLOCKS if independent_blocks else 0, 0, index_groups.ctypes.data if index_groups is not None else None,
                      index_groups.size if index_groups is not None else 0)
    m = L.hsrans_encode_ex(container, states, bits, _p(data), data.size, _p(out), cap, hp, ctypes.byref(opts))
    if m == 0:
        raise HsransError("encode failed")
    if not want_plan:
        return out[:m].copy()
    return out[:m].copy(), plan[:opts.plan_size].copy()


def plan_thin(plan, groups) -> np.ndarray:
    plan = _u8(plan)
    groups = np.ascontiguousarray(groups, dtype=np.uint64)
    out = np.zeros(plan.size, np.uint8)
    n = load_library().hsrans_plan_thin(_p(plan), plan.size, _p(groups), groups.size, _p(out), out.size)
    if n == 0:
        raise HsransError("plan_thin failed")
    return out[:n].copy()


def plan_build(container: int, states: int, bits: int, stream, out_capacity: int | None = None) -> np.ndarray:
    L = load_library()
    stream = _u8(stream)
    if stream.size < 16:
        raise HsransError("stream too short")
    out_len = int(stream[:8].view(np.uint64)[0])
    if out_capacity is None:
        out_capacity = out_len
    pcap = L.hsrans_plan_capacity(container, states, min(out_len, 1 << 40), 0, 0)
    plan = np.zeros(pcap, np.uint8)
    n = L.hsrans_plan_build(container, states, bits, _p(stream), stream.size, out_capacity, _p(plan), pcap)
    if n == 0 and container != RAW:
        # blocks below the 32 KiB hsrans_plan_capacity assumes for foreign streams (encode(block_size=...)): one chain per
        # block header of the stream at most
        pcap = min(L.hsrans_plan_capacity(container, states, min(out_len, 1 << 40), 0, 64), stream.size * 40 + (1 << 20))
        plan = np.zeros(pcap, np.uint8)
        n = L.hsrans_plan_build(container, states, bits, _p(stream), stream.size, out_capacity, _p(plan), pcap)
    if n == 0:
        raise HsransError("malformed stream (plan_build returned 0)")
    return plan[:n].copy()


def plan_chain_count(plan) -> int:
    plan = _u8(plan)
    return load_library().hsrans_plan_chain_count(_p(plan), plan.size)


def plan_decoded_length(plan) -> int:
    plan = _u8(plan)
    return load_library().hsrans_plan_decoded_length(_p(plan), plan.size)


def plan_slice(plan, first: int, count: int) -> np.ndarray:
    plan = _u8(plan)
    out = np.zeros(plan.size, np.uint8)
    n = load_library().hsrans_plan_slice(_p(plan), plan.size, first, count, _p(out), out.size)
    if n == 0:
        raise HsransError("plan_slice failed")
    return out[:n].copy()


def plan_chain_range(plan, first: int, count: int) -> tuple[int, int]:
    plan = _u8(plan)
    b, e = ctypes.c_uint64(), ctypes.c_uint64()
    if load_library().hsrans_plan_chain_range(_p(plan), plan.size, first, count, ctypes.byref(b), ctypes.byref(e)) != 0:
        raise HsransError("plan_chain_range failed")
    return b.value, e.value


def plan_stream_ranges(plan, first: int, count: int) -> tuple[tuple[int, int], tuple[int, int]]:
    """((head_begin, head_end), (body_begin, body_end)): the stream bytes chains [first, first+count) can read."""
    plan = _u8(plan)
    r = (ctypes.c_uint64 * 4)()
    if load_library().hsrans_plan_stream_ranges(_p(plan), plan.size, first, count, r) != 0:
        raise HsransError("plan_stream_ranges failed")
    return (r[0], r[1]), (r[2], r[3])


# ---------------------------------------------------------------------------------------------------------------------
# host SIMD decoders (csrc/hsrans_cpu.cpp): the CPU comparator / single-chain path; never used by Context (the GPU entries)
# ---------------------------------------------------------------------------------------------------------------------
CPU_LEVELS = {0: "scalar", 1: "avx2", 2: "avx512"}


def cpu_level() -> int:
    return load_library().hsrans_cpu_level()


def decode_cpu(container: int, states: int, bits: int, stream, out_capacity: int | None = None, plan=None, level: int = -1, threads: int = 1,
               in_length: int | None = None):
    """Returns (returned_length, out) like the reference's decoders (0 = failure).  hsrans_decode_cpu."""
    stream = _u8(stream)
    if out_capacity is None:
        out_capacity = int(stream[:8].view(np.uint64)[0]) if stream.size >= 8 else 0
    out = np.full(max(out_capacity, 1) + 64, 0xCC, np.uint8)
    pp, pn = (None, 0)
    if plan is not None:
        plan = _u8(plan)
        pp, pn = _p(plan), plan.size
    r = load_library().hsrans_decode_cpu(level, threads, container, states, bits, _p(stream), stream.size if in_length is None else in_length, _p(out),
                                         out_capacity, pp, pn)
    return r, out[:out_capacity]


def index_build_host(container: int, states: int, bits: int, stream, groups, level: int = -1, threads: int = 1) -> np.ndarray:
    stream = _u8(stream)
    groups = np.ascontiguousarray(groups, dtype=np.uint64)
    out_len = int(stream[:8].view(np.uint64)[0])
    L = load_library()
    pcap = L.hsrans_plan_capacity_chains(container, states, out_len, groups.size, 0)
    plan = np.zeros(pcap, np.uint8)
    n = L.hsrans_index_build_host(level, threads, container, states, bits, _p(stream), stream.size, _p(groups), groups.size, _p(plan), pcap)
    if n == 0:
        raise HsransError("hsrans_index_build_host failed")
    return plan[:n].copy()


PIECE_DTYPE = np.dtype([("words_off", "<u8"), ("out_off", "<u8"), ("hist_off", "<u8"), ("fill_len", "<u8"), ("steps", "<u4"),
                        ("tail", "<u2"), ("flags", "<u2"), ("state_idx", "<u4"), ("reserved", "<u4")])


def plan_tables(plan):
    """Read-only numpy views of a plan blob (layout: csrc/hsrans_plan.h): (header dict, chain_first[n+1], pieces[n_pieces])."""
    plan = _u8(plan)
    if plan.size < 64 or bytes(plan[:8]) != b"HSRPLAN1":
        raise HsransError("not a plan blob")
    u32 = plan[8:24].view("<u4")
    hdr = {"container": int(u32[0]), "states": int(u32[1]), "bits": int(u32[2]), "flags": int(u32[3]),
           "decoded_len": int(plan[24:32].view("<u8")[0]), "stream_len": int(plan[32:40].view("<u8")[0]),
           "n_chains": int(plan[40:44].view("<u4")[0]), "n_pieces": int(plan[44:48].view("<u4")[0]),
           "shared_hist": int(plan[48:52].view("<u4")[0]), "interval": int(plan[52:56].view("<u4")[0])}
    nc, npc = hdr["n_chains"], hdr["n_pieces"]
    cf = plan[64:64 + 4 * (nc + 1)].view("<u4")
    po = 64 + ((nc + 1) * 4 + 15) // 16 * 16
    pieces = plan[po:po + 48 * npc].view(PIECE_DTYPE)
    return hdr, cf, pieces


def dealt_shares(block_begin, total_groups: int, ctx: "Context | None" = None, bits: int = 11):
    """hsrans_dealt_shares: (suits the launch?, begin[513], split[512]) for a plan whose block k is chains [block_begin[k], block_begin[k + 1])"""
    bb = np.ascontiguousarray(block_begin, dtype=np.uint32)
    begin, split = np.zeros(513, np.uint32), np.zeros(512, np.uint16)
    rc = load_library().hsrans_dealt_shares(ctx.handle if ctx is not None else None, bits, _p(bb), bb.size - 1, int(bb[-1]), int(total_groups), _p(begin), _p(split))
    if rc < 0:
        raise HsransError("hsrans_dealt_shares: bad arguments")
    return rc == 1, begin, split


def index_boundaries_batch(states: int, bits: int, decoded_sizes, member: int, ctx: "Context | None" = None) -> np.ndarray:
    """hsrans_index_boundaries_batch: checkpoint positions for member ``member`` of a batch of streams of ``decoded_sizes``."""
    sizes = (ctypes.c_size_t * len(decoded_sizes))(*[int(x) for x in decoded_sizes])
    out = np.zeros(1 << 15, np.uint64)
    n = load_library().hsrans_index_boundaries_batch(ctx.handle if ctx is not None else None, states, bits, sizes, len(decoded_sizes), member, _p(out), out.size)
    return out[:n].copy()


def shard_layout(plan, world: int, parts: int = 1, weights=None):
    """hsrans_shard_layout: (shards[world][parts] as (first_chain, chain_count, out_begin, out_end), windows[world] as (begin, end)).
    Pure host arithmetic in the C library — what every rank of a sharded decode computes identically."""
    plan = _u8(plan)
    shards = (Shard * (world * parts))()
    windows = np.zeros(2 * world, np.uint64)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    if w is not None and w.size != world:
        raise HsransError("shard_layout: one weight per rank")
    rc = load_library().hsrans_shard_layout(_p(plan), plan.size, world, parts, None if w is None else _p(w), shards, _p(windows))
    if rc != 0:
        raise HsransError(f"hsrans_shard_layout failed with code {rc}")
    out = [[(int(shards[r * parts + k].first_chain), int(shards[r * parts + k].chain_count), int(shards[r * parts + k].out_begin), int(shards[r * parts + k].out_end))
            for k in range(parts)] for r in range(world)]
    return out, [(int(windows[2 * r]), int(windows[2 * r + 1])) for r in range(world)]


def batch_deal(chain_starts, grid: int = 512, waves: int = 16, weights=None):
    """hsrans_batch_deal: how one launch's wave slots would be dealt to members whose chains start at ``chain_starts[m]`` (groups,
    ascending, last entry = the member's total).  Returns (imbalance, slots[grid * waves, 4] = member, first chain, end chain, flags)."""
    L = load_library()
    arrs = [np.ascontiguousarray(c, dtype=np.uint64) for c in chain_starts]
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    n_chains = np.array([a.size - 1 for a in arrs], np.uint32)
    slots = np.zeros(grid * waves * 4, np.uint32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
    imb = L.hsrans_batch_deal(ptrs, _p(n_chains), len(arrs), grid, waves, None if w is None else _p(w), _p(slots))
    if imb < 0:
        raise HsransError("hsrans_batch_deal: bad arguments")
    return float(imb), slots.reshape(-1, 4)


# ---------------------------------------------------------------------------------------------------------------------
# GPU side
# ---------------------------------------------------------------------------------------------------------------------
class Batch:
    """hsrans_batch: K device plans decoded by one launch (Context.make_batch / Context.decode_device_batch)."""

    def __init__(self, ctx: "Context", handle, dplans):
        self.ctx, self.handle, self.dplans = ctx, handle, list(dplans)  # (keeps the member plans alive)

    def info(self) -> dict:
        info = BatchInfo()
        load_library().hsrans_dplan_batch_info(self.handle, ctypes.byref(info))
        return {n: (list(getattr(info, n)) if n == "class_weights" else getattr(info, n)) for n, _ in BatchInfo._fields_}

    def read_finish(self) -> np.ndarray:
        out = np.zeros(1 << 16, np.uint64)
        n = load_library().hsrans_dplan_batch_read_finish(self.handle, _p(out), out.size)
        return out[:n].copy()

    def close(self):
        if self.handle:
            load_library().hsrans_dplan_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePlan:
    def __init__(self, ctx: "Context", handle, owned: bool = True):
        self.ctx, self.handle, self.owned = ctx, handle, owned  # (owned False: a view of a plan another object destroys, e.g. a Sharded's sub-run)

    def launch_info(self) -> dict:
        info = LaunchInfo()
        load_library().hsrans_dplan_launch_info(self.handle, ctypes.byref(info))
        return {n: (list(getattr(info, n)) if n == "class_weights" else getattr(info, n)) for n, _ in LaunchInfo._fields_}

    def close(self):
        if self.handle and self.owned:
            load_library().hsrans_dplan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """hsrans_comm: this process's rank in a communicator over the GPUs of the node (RCCL, bound by the C library at run time)."""

    def __init__(self, ctx: "Context", comm_id: bytes, rank: int, world: int):
        assert len(comm_id) == COMM_ID_BYTES
        self.ctx = ctx
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
        h = _vp()
        rc = ctx.L.hsrans_comm_create(ctx.handle, buf, rank, world, ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_comm_create(rank {rank} of {world}) failed with code {rc}")
        self.handle, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
        rc = load_library().hsrans_comm_unique_id(buf)
        if rc != 0:
            raise HsransError(f"hsrans_comm_unique_id failed with code {rc} (no RCCL could be bound?)")
        return bytes(buf)

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.L.hsrans_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Queue:
    """hsrans_queue: streams submitted one by one, decoded by ONE batch launch per flush (the dealing cached per shape)."""

    def __init__(self, ctx: "Context", max_members: int = 32):
        self.ctx = ctx
        h = _vp()
        rc = ctx.L.hsrans_queue_create(ctx.handle, max_members, ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_queue_create failed with code {rc}")
        self.handle = h
        self._keep = []  # tensors and plans of the submissions still pending

    def submit(self, dplan: "DevicePlan", d_stream: torch.Tensor, d_out: torch.Tensor, stream_length: int | None = None, stream: torch.cuda.Stream | None = None):
        s = stream if stream is not None else torch.cuda.current_stream(d_out.device)
        rc = self.ctx.L.hsrans_queue_submit(self.handle, dplan.handle, d_stream.data_ptr(), d_stream.numel() if stream_length is None else stream_length,
                                            d_out.data_ptr(), d_out.numel(), ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_queue_submit failed with code {rc}")
        self._keep.append((dplan, d_stream, d_out))
        if self.pending() == 0:
            self._keep.clear()

    def flush(self, stream: torch.cuda.Stream | None = None):
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = self.ctx.L.hsrans_queue_flush(self.handle, ctypes.c_void_p(s.cuda_stream))
        self._keep.clear()
        if rc != 0:
            raise HsransError(f"hsrans_queue_flush failed with code {rc}")

    def pending(self) -> int:
        return int(self.ctx.L.hsrans_queue_pending(self.handle))

    def stats(self) -> dict:
        st = QueueStats()
        self.ctx.L.hsrans_queue_stats(self.handle, ctypes.byref(st))
        return {n: int(getattr(st, n)) for n, _ in QueueStats._fields_}

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.L.hsrans_queue_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sharded:
    """hsrans_sharded: this rank's share of ONE stream decoded by all ranks of a communicator (or, without one, decode only)."""

    def __init__(self, ctx: "Context", plan, parts: int = 1, weights=None, root: int | None = None, comm: Comm | None = None, rank: int | None = None,
                 world: int | None = None):
        plan = _u8(plan)
        self.ctx, self.comm = ctx, comm
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        h = _vp()
        r = -1 if root is None else int(root)
        if comm is not None:
            rc = ctx.L.hsrans_sharded_create(ctx.handle, comm.handle, _p(plan), plan.size, parts, None if w is None else _p(w), r, ctypes.byref(h))
        else:
            rc = ctx.L.hsrans_sharded_create_rank(ctx.handle, rank, world, _p(plan), plan.size, parts, None if w is None else _p(w), r, ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_sharded_create failed with code {rc}")
        self.handle = h
        info = ShardedInfo()
        shards = (Shard * (int(world if comm is None else comm.world) * parts))()
        ctx.L.hsrans_sharded_info(h, ctypes.byref(info), shards, len(shards))
        self.info = {n: getattr(info, n) for n, _ in ShardedInfo._fields_}
        self.shards = [[(int(shards[q * parts + k].first_chain), int(shards[q * parts + k].chain_count), int(shards[q * parts + k].out_begin), int(shards[q * parts + k].out_end))
                        for k in range(parts)] for q in range(info.world)]

    def part_plan(self, k: int) -> DevicePlan | None:
        h = self.ctx.L.hsrans_sharded_part_plan(self.handle, k)
        return DevicePlan(self.ctx, _vp(h), owned=False) if h else None

    def wait_part(self, part: int, stream: torch.cuda.Stream):
        """hsrans_sharded_wait_part: `stream` waits until sub-run `part` of the last decode() is complete and visible"""
        rc = self.ctx.L.hsrans_sharded_wait_part(self.handle, part, ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_sharded_wait_part failed with code {rc}")

    def whole_plan(self) -> DevicePlan | None:
        """this rank's whole run when ONE launch decodes all its sub-runs (hsrans_sharded_info: one_launch), else None"""
        h = self.ctx.L.hsrans_sharded_whole_plan(self.handle)
        return DevicePlan(self.ctx, _vp(h), owned=False) if h else None

    def decode(self, d_window: torch.Tensor, d_out: torch.Tensor, mode: int = SHARD_DECODE_AND_EXCHANGE, stream: torch.cuda.Stream | None = None):
        s = stream if stream is not None else torch.cuda.current_stream(d_out.device)
        rc = self.ctx.L.hsrans_decode_sharded(self.handle, d_window.data_ptr(), d_out.data_ptr(), mode, ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_decode_sharded failed with code {rc}")

    def status(self, stream: torch.cuda.Stream | None = None) -> int:
        s = stream if stream is not None else torch.cuda.current_stream()
        return self.ctx.L.hsrans_sharded_status(self.handle, ctypes.c_void_p(s.cuda_stream))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.L.hsrans_sharded_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """hsrans_ctx: one per GPU.  Raises if no gfx950 device is usable (there is no CPU decode path)."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = _vp()
        rc = self.L.hsrans_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_ctx_create(device={device}) failed with code {rc}: a gfx950 GPU is required")
        self.handle = h
        self.device = device

    @property
    def device_name(self) -> str:
        return self.L.hsrans_ctx_device_name(self.handle).decode()

    def close(self):
        if getattr(self, "handle", None):
            self.L.hsrans_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def calibrate(self, bits: int = 11, iterations: int = 0) -> dict:
        """hsrans_ctx_calibrate: fits the one-chain-per-wave index's class lengths to this device; returns the report."""
        rep = Calibration()
        rc = self.L.hsrans_ctx_calibrate(self.handle, bits, iterations, ctypes.byref(rep))
        if rc != 0:
            raise HsransError(f"hsrans_ctx_calibrate failed with code {rc}")
        return {"class_weights": list(rep.class_weights), "class_finish_us_last_iteration": [round(v, 3) for v in rep.class_finish_us_last_iteration],
                "last_wave_us_before": rep.last_wave_us_before, "last_wave_us_after": rep.last_wave_us_after,
                "class_spread_us_before": rep.class_spread_us_before, "class_spread_us_after": rep.class_spread_us_after,
                "iterations": rep.iterations, "bytes": rep.bytes}

    def calibrate_runs(self, bits: int = 11, copies: int = 2, iterations: int = 0) -> dict:
        """hsrans_ctx_calibrate_runs: the same fit for runs ``copies`` times as long (``copies`` members of the calibration stream in one launch)."""
        rep = Calibration()
        rc = self.L.hsrans_ctx_calibrate_runs(self.handle, bits, iterations, copies, ctypes.byref(rep))
        if rc != 0:
            raise HsransError(f"hsrans_ctx_calibrate_runs failed with code {rc}")
        return {"copies": copies, "class_weights": list(rep.class_weights), "class_finish_us_last_iteration": [round(v, 3) for v in rep.class_finish_us_last_iteration],
                "last_wave_us_before": rep.last_wave_us_before, "last_wave_us_after": rep.last_wave_us_after,
                "class_spread_us_before": rep.class_spread_us_before, "class_spread_us_after": rep.class_spread_us_after,
                "iterations": rep.iterations, "bytes": rep.bytes}

    # -- host-pointer drop-in: decodeFunc(pInData, inLength, pOutData, outCapacity) ------------------------------
    def decode_host(self, container: int, states: int, bits: int, stream, out_capacity: int, plan=None, in_length: int | None = None):
        """Returns (returned_length, out) like the reference's decoders: returned_length == 0 means failure."""
        stream = _u8(stream)
        out = np.full(max(out_capacity, 1), 0xCC, np.uint8)
        pp, pn = (None, 0)
        if plan is not None:
            plan = _u8(plan)
            pp, pn = _p(plan), plan.size
        r = self.L.hsrans_decode_host(self.handle, container, states, bits, _p(stream), stream.size if in_length is None else in_length, _p(out),
                                      out_capacity, pp, pn)
        return r, out[:out_capacity]

    def host_index_chains(self) -> int:
        """chains of the index decode_host keeps from its last plan-less call on an mt_/raw stream (0 = none)"""
        return int(self.L.hsrans_ctx_host_index_chains(self.handle))

    def decode(self, container: int, states: int, bits: int, stream, plan=None) -> np.ndarray:
        stream = _u8(stream)
        n = int(stream[:8].view(np.uint64)[0]) if stream.size >= 8 else 0
        r, out = self.decode_host(container, states, bits, stream, n, plan)
        if r == 0:
            raise HsransError("decode failed (malformed stream or no device)")
        return out[:r]

    # -- device-resident path ------------------------------------------------------------------------------------
    def make_device_plan(self, plan) -> DevicePlan:
        plan = _u8(plan)
        h = _vp()
        rc = self.L.hsrans_dplan_create(self.handle, _p(plan), plan.size, ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_dplan_create failed with code {rc}")
        return DevicePlan(self, h)

    def make_device_plan_from_stream(self, container: int, states: int, bits: int, d_stream: torch.Tensor, stream_length: int,
                                     out_capacity: int, stream: torch.cuda.Stream | None = None) -> DevicePlan:
        """mt_ only: the header chain is walked on the GPU (no host copy of the stream needed)."""
        s = stream if stream is not None else torch.cuda.current_stream(d_stream.device)
        h = _vp()
        rc = self.L.hsrans_dplan_create_from_device_stream(self.handle, container, states, bits, d_stream.data_ptr(), stream_length, out_capacity,
                                                           ctypes.c_void_p(s.cuda_stream), ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_dplan_create_from_device_stream failed with code {rc}")
        return DevicePlan(self, h)

    def read_device_plan(self, dplan: DevicePlan, capacity: int = 1 << 28) -> np.ndarray:
        out = np.zeros(capacity, np.uint8)
        n = self.L.hsrans_dplan_read_plan(dplan.handle, _p(out), out.size)
        if n == 0:
            raise HsransError("hsrans_dplan_read_plan failed")
        return out[:n].copy()

    def decode_device(self, dplan: DevicePlan, d_stream: torch.Tensor, d_out: torch.Tensor, stream: torch.cuda.Stream | None = None,
                      stream_length: int | None = None):
        """Asynchronous launch on ``stream`` (default: torch's current stream).  Tensors are uint8, on this context's GPU."""
        s = stream if stream is not None else torch.cuda.current_stream(d_stream.device)
        rc = self.L.hsrans_decode_device(self.handle, dplan.handle, d_stream.data_ptr(), d_stream.numel() if stream_length is None else stream_length,
                                         d_out.data_ptr(), d_out.numel(), ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_decode_device failed with code {rc}")

    def decode_device_indexing(self, dplan: DevicePlan, d_stream: torch.Tensor, d_out: torch.Tensor, index_interval: int,
                               stream: torch.cuda.Stream | None = None, stream_length: int | None = None) -> DevicePlan:
        """First decode of a stream without an index: fills ``d_out`` and returns the plan with a checkpoint every
        ``index_interval`` groups for the later decodes (hsrans_decode_device_indexing; synchronises ``stream``)."""
        s = stream if stream is not None else torch.cuda.current_stream(d_stream.device)
        h = _vp()
        rc = self.L.hsrans_decode_device_indexing(self.handle, dplan.handle, d_stream.data_ptr(), d_stream.numel() if stream_length is None else stream_length,
                                                  d_out.data_ptr(), d_out.numel(), index_interval, ctypes.c_void_p(s.cuda_stream), ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_decode_device_indexing failed with code {rc}")
        return DevicePlan(self, h)

    def decode_device_window(self, dplan: DevicePlan, d_window: torch.Tensor, window_offset: int, window_length: int, d_out: torch.Tensor,
                             stream: torch.cuda.Stream | None = None):
        """As decode_device for a caller that holds only stream bytes [window_offset, window_offset + window_length) (hsrans_decode_device_window)."""
        s = stream if stream is not None else torch.cuda.current_stream(d_window.device)
        rc = self.L.hsrans_decode_device_window(self.handle, dplan.handle, d_window.data_ptr(), window_offset, window_length, d_out.data_ptr(), d_out.numel(),
                                                ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_decode_device_window failed with code {rc}")

    def decode_device_ranges(self, dplan: DevicePlan, d_window: torch.Tensor, window_offset: int, window_length: int, d_out_window: torch.Tensor,
                             out_offset: int, out_length: int, stream: torch.cuda.Stream | None = None):
        """As decode_device_window for a caller that also holds only output bytes [out_offset, out_offset + out_length)
        (hsrans_decode_device_ranges): one rank's share of a sharded decode."""
        s = stream if stream is not None else torch.cuda.current_stream(d_window.device)
        assert d_out_window.numel() >= out_length
        rc = self.L.hsrans_decode_device_ranges(self.handle, dplan.handle, d_window.data_ptr(), window_offset, window_length, d_out_window.data_ptr(),
                                                out_offset, out_length, ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_decode_device_ranges failed with code {rc}")

    # -- K independent streams, one launch -------------------------------------------------------------------------
    def make_batch(self, dplans) -> Batch:
        dplans = list(dplans)
        arr = (ctypes.c_void_p * len(dplans))(*[d.handle.value if isinstance(d.handle, ctypes.c_void_p) else d.handle for d in dplans])
        h = _vp()
        rc = self.L.hsrans_dplan_batch_create(self.handle, arr, len(dplans), ctypes.byref(h))
        if rc != 0:
            raise HsransError(f"hsrans_dplan_batch_create failed with code {rc}")
        return Batch(self, h, dplans)

    def decode_device_batch(self, batch: Batch, d_streams, d_outs, stream: torch.cuda.Stream | None = None, stream_lengths=None):
        """Asynchronous on ``stream`` (default: torch's current stream): member k decodes d_streams[k] into d_outs[k]."""
        K = len(batch.dplans)
        assert len(d_streams) == K and len(d_outs) == K
        s = stream if stream is not None else torch.cuda.current_stream(d_streams[0].device)
        sp = (ctypes.c_void_p * K)(*[t.data_ptr() for t in d_streams])
        op = (ctypes.c_void_p * K)(*[t.data_ptr() for t in d_outs])
        sl = (ctypes.c_size_t * K)(*[(t.numel() if stream_lengths is None else int(stream_lengths[k])) for k, t in enumerate(d_streams)])
        oc = (ctypes.c_size_t * K)(*[t.numel() for t in d_outs])
        rc = self.L.hsrans_decode_device_batch(self.handle, batch.handle, sp, sl, op, oc, ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise HsransError(f"hsrans_decode_device_batch failed with code {rc}")

    def batch_status(self, batch: Batch, stream: torch.cuda.Stream | None = None) -> list:
        s = stream if stream is not None else torch.cuda.current_stream()
        codes = (ctypes.c_int * len(batch.dplans))()
        self.L.hsrans_dplan_batch_status(self.handle, batch.handle, ctypes.c_void_p(s.cuda_stream), codes)
        return list(codes)

    def status(self, dplan: DevicePlan, stream: torch.cuda.Stream | None = None) -> int:
        s = stream if stream is not None else torch.cuda.current_stream()
        return self.L.hsrans_dplan_status(self.handle, dplan.handle, ctypes.c_void_p(s.cuda_stream))

    def encode_device(self, container: int, states: int, bits: int, d_in: torch.Tensor, d_out: torch.Tensor, block_size: int = 1 << 16,
                      index_interval: int = 0, want_plan: bool = False, stream: torch.cuda.Stream | None = None):
        """GPU encoder (mt_, independent fixed-size blocks).  Returns the stream length written to d_out, or
        ``(length, DevicePlan)`` with ``want_plan`` (plan with a checkpoint every ``index_interval`` groups, built on the device)."""
        s = stream if stream is not None else torch.cuda.current_stream(d_in.device)
        h = _vp()
        n = self.L.hsrans_encode_device(self.handle, container, states, bits, d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(), block_size,
                                        index_interval, ctypes.c_void_p(s.cuda_stream), ctypes.byref(h) if want_plan else None)
        if n == 0:
            raise HsransError("hsrans_encode_device failed")
        return (n, DevicePlan(self, h)) if want_plan else n

    def encode_device_raw(self, states: int, bits: int, d_in: torch.Tensor, d_out: torch.Tensor, hist=None, index_interval: int = 0, index_groups=None,
                          want_plan: bool = False, want_device_plan: bool = False, stream: torch.cuda.Stream | None = None):
        """GPU encoder for the raw format (one wavefront: hsrans_encode_device_raw).  Returns the stream length, followed by the plan
        blob (``want_plan``) and/or a DevicePlan (``want_device_plan``) when an index (``index_interval`` or ``index_groups``) is asked for."""
        s = stream if stream is not None else torch.cuda.current_stream(d_in.device)
        groups = np.ascontiguousarray(index_groups, dtype=np.uint64) if index_groups is not None else None
        n_groups = 0 if groups is None else groups.size
        plan, psize, h = None, _sz(0), _vp()
        if want_plan:
            pcap = (self.L.hsrans_plan_capacity_chains(RAW, states, d_in.numel(), n_groups, 0) if n_groups
                    else self.L.hsrans_plan_capacity(RAW, states, d_in.numel(), index_interval, 0))
            plan = np.zeros(pcap, np.uint8)
        n = self.L.hsrans_encode_device_raw(self.handle, states, bits, d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(),
                                            ctypes.addressof(hist) if hist is not None else None, index_interval, _p(groups) if n_groups else None, n_groups,
                                            _p(plan) if want_plan else None, plan.size if want_plan else 0, ctypes.byref(psize),
                                            ctypes.c_void_p(s.cuda_stream), ctypes.byref(h) if want_device_plan else None)
        if n == 0:
            raise HsransError("hsrans_encode_device_raw failed")
        out = [n]
        if want_plan:
            out.append(plan[: psize.value].copy())
        if want_device_plan:
            out.append(DevicePlan(self, h))
        return out[0] if len(out) == 1 else tuple(out)

    def index_build_at(self, container: int, states: int, bits: int, stream, groups) -> np.ndarray:
        stream = _u8(stream)
        groups = np.ascontiguousarray(groups, dtype=np.uint64)
        out_len = int(stream[:8].view(np.uint64)[0])
        pcap = self.L.hsrans_plan_capacity_chains(container, states, out_len, groups.size, 0)
        plan = np.zeros(pcap, np.uint8)
        n = self.L.hsrans_index_build_at(self.handle, container, states, bits, _p(stream), stream.size, _p(groups), groups.size, _p(plan), pcap)
        if n == 0:
            raise HsransError("hsrans_index_build_at failed")
        return plan[:n].copy()

    def decode_host_pipelined(self, container: int, states: int, bits: int, stream: torch.Tensor, out: torch.Tensor, plan, n_slices: int = 0) -> int:
        """Host tensors (ideally pinned); returns the decoded length (0 = failure).  hsrans_decode_host_pipelined."""
        plan = _u8(plan)
        return self.L.hsrans_decode_host_pipelined(self.handle, container, states, bits, stream.data_ptr(), stream.numel(), out.data_ptr(), out.numel(),
                                                   _p(plan), plan.size, n_slices)

    def index_build(self, container: int, states: int, bits: int, stream, index_interval: int) -> np.ndarray:
        stream = _u8(stream)
        out_len = int(stream[:8].view(np.uint64)[0])
        pcap = self.L.hsrans_plan_capacity(container, states, out_len, index_interval, 0)
        plan = np.zeros(pcap, np.uint8)
        n = self.L.hsrans_index_build(self.handle, container, states, bits, _p(stream), stream.size, index_interval, _p(plan), pcap)
        if n == 0:
            raise HsransError("hsrans_index_build failed")
        return plan[:n].copy()
