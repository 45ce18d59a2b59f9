"""hypersonic_rans_amd — MI355X (gfx950) implementation of hypersonic-rANS's interleaved 32-bit-state / 16-bit-word
rANS decode path (rANS32x32 16w / rANS32x64 16w; raw, block_ and mt_ containers; 10..15-bit histograms).

The product is the C-ABI library ``lib/libhsrans_hip.so`` (``include/hsrans_hip.h``) holding hand-written HIP kernels;
this package is the thin host-side mirror of the reference's codec interface on top of it (ctypes; torch is used only for
device buffers, streams and torch.distributed).  There is no CPU decode path: decoding raises without a gfx950 device.
"""
from .api import (  # noqa: F401
    BLOCK,
    MT,
    RAW,
    Context,
    HsransError,
    capacity,
    encode,
    lib_path,
    load_library,
    make_hist,
    plan_build,
    plan_chain_count,
    plan_chain_range,
    plan_decoded_length,
    plan_slice,
    plan_stream_ranges,
    plan_thin,
    index_boundaries,
    batch_deal,
    index_boundaries_batch,
)

__all__ = [
    "RAW", "BLOCK", "MT", "Context", "HsransError", "capacity", "encode", "make_hist", "plan_build", "plan_chain_count",
    "plan_chain_range", "plan_decoded_length", "plan_slice", "plan_stream_ranges", "plan_thin", "index_boundaries", "batch_deal", "index_boundaries_batch", "lib_path", "load_library",
]
