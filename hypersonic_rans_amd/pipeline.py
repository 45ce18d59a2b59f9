"""Host-resident streams: upload, decode and download overlapped on separate HIP streams (BASELINE config 5 shape).

The reference decodes host buffers in place (mt_rANS32x64_16w_decode.cpp:137-265: a thread pool over the blocks).  With the
stream and the output in (pinned) host memory the GPU version is PCIe-bound, so the three legs are pipelined over slices of
the plan's chains: slice k's compressed bytes go up on one stream while slice k-1 decodes on another and slice k-2's decoded
bytes come down on a third — the two DMA directions of the link run concurrently and the kernels hide behind them.

The pipeline itself lives behind the C ABI (hsrans_hpipe_* in include/hsrans_hip.h, csrc/hsrans_capi.cpp: three HIP streams,
slice plans resident on the device, events between the legs) so that any host language gets it; this module is the thin
Python caller.
"""
from __future__ import annotations

import ctypes

import torch

from . import api


class PipelinedHostDecoder:
    """hsrans_hpipe: prepared once per (plan, slicing); `decode` can then be called repeatedly for streams that share the
    plan's layout (in practice: the same stream).  `host_stream` and `host_out` should be pinned (torch `pin_memory()`,
    or hsrans_host_register): with pageable memory the runtime stages the copies itself and the legs no longer overlap."""

    def __init__(self, ctx: "api.Context", plan, n_slices: int = 0, device: "torch.device | None" = None):
        self.ctx = ctx
        self.L = api.load_library()
        plan = api._u8(plan)
        self.total = api.plan_decoded_length(plan)
        h = ctypes.c_void_p()
        rc = self.L.hsrans_hpipe_create(ctx.handle, api._p(plan), plan.size, n_slices, ctypes.byref(h))
        if rc != 0:
            raise api.HsransError(f"hsrans_hpipe_create failed with code {rc}")
        self.handle = h

    def decode(self, host_stream: torch.Tensor, host_out: torch.Tensor) -> None:
        """Synchronous: `host_out[:total]` holds the decoded bytes on return."""
        assert host_stream.dtype == torch.uint8 and host_out.dtype == torch.uint8 and host_out.numel() >= self.total
        assert not host_stream.is_cuda and not host_out.is_cuda
        r = self.L.hsrans_hpipe_decode(self.handle, host_stream.data_ptr(), host_stream.numel(), host_out.data_ptr(), host_out.numel())
        if r != self.total:
            raise api.HsransError("hsrans_hpipe_decode failed (malformed stream / plan mismatch)")

    def close(self):
        if getattr(self, "handle", None):
            self.L.hsrans_hpipe_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_from_host_unpipelined(ctx: "api.Context", plan, host_stream: torch.Tensor, host_out: torch.Tensor, device=None) -> None:
    """The same work without overlap (upload everything, decode, download everything): the comparison point."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    total = api.plan_decoded_length(plan)
    n = host_stream.numel()
    d_stream = torch.empty(n + (-n) % 16, dtype=torch.uint8, device=device)
    d_stream[:n].copy_(host_stream, non_blocking=True)
    d_out = torch.empty(total, dtype=torch.uint8, device=device)
    dplan = ctx.make_device_plan(plan)
    ctx.decode_device(dplan, d_stream, d_out, stream_length=n)
    host_out[:total].copy_(d_out, non_blocking=True)
    torch.cuda.synchronize(device)
    if ctx.status(dplan) != 0:
        raise api.HsransError("device reported a malformed histogram / block header")
